"""Push streaming (include/zxc_pstream.h) without a GPU: the host logic of the two state machines — argument checks, refusals,
the framing of an empty stream, header errors and their stickiness, bytes behind the footer — against the UNMODIFIED reference
driven through the same prototypes; and that a block to process fails loudly (ZXC_ERROR_GPU_UNAVAILABLE, sticky) instead of
falling back to anything. The parity tests proper (data blocks) are tests/test_gpu_pstream.py."""
import ctypes as C

import pytest

import zxc_amd.api as api

NULL_INPUT, GPU_UNAVAILABLE, BAD_MAGIC, BAD_HEADER, BAD_BLOCK_SIZE, CORRUPT = -12, -100, -4, -6, -14, -8


@pytest.fixture(scope="module")
def L(product):
    return api._bind_pstream(product.lib())


def _no_gpu(product):
    return product.lib().zxc_mi355x_device_count() <= 0


def test_null_and_malformed_arguments(L):
    """reference tests/test_pstream_api.c:389-459 (test_pstream_invalid_args)"""
    assert L.zxc_cstream_compress(None, None, None) == NULL_INPUT
    assert L.zxc_cstream_end(None, None) == NULL_INPUT
    assert L.zxc_dstream_decompress(None, None, None) == NULL_INPUT
    for f in (L.zxc_cstream_in_size, L.zxc_cstream_out_size, L.zxc_dstream_in_size, L.zxc_dstream_out_size):
        assert f(None) == 0
    assert L.zxc_dstream_finished(None) == 0
    L.zxc_cstream_free(None)
    L.zxc_dstream_free(None)
    ds = L.zxc_dstream_create(None)
    assert ds
    buf = C.create_string_buffer(16)
    good_out = api._OutBuf(C.addressof(buf), 16, 0)
    empty_in = api._InBuf(None, 0, 0)
    for out, inb in ((good_out, api._InBuf(C.addressof(buf), 4, 5)),      # in.pos > in.size
                     (api._OutBuf(C.addressof(buf), 4, 5), empty_in),     # out.pos > out.size
                     (good_out, api._InBuf(None, 16, 0)),                 # bytes claimed, src NULL
                     (api._OutBuf(None, 16, 0), empty_in)):               # capacity claimed, dst NULL
        assert L.zxc_dstream_decompress(ds, C.byref(out), C.byref(inb)) == NULL_INPUT
    L.zxc_dstream_free(ds)
    cs = L.zxc_cstream_create(None)
    assert cs
    assert L.zxc_cstream_compress(cs, C.byref(good_out), C.byref(api._InBuf(None, 16, 0))) == NULL_INPUT
    assert L.zxc_cstream_in_size(cs) >= 512 * 1024 and L.zxc_cstream_out_size(cs) > L.zxc_cstream_in_size(cs)
    L.zxc_cstream_free(cs)


def test_dictionaries_and_bad_block_sizes_are_refused(L):
    d = C.create_string_buffer(b"\x01\x02\x03\x04", 4)
    co = api._CompressOpts(level=3, dict=C.cast(d, C.c_void_p), dict_size=4)
    assert not L.zxc_cstream_create(C.byref(co))
    do = api._DecompressOpts(dict=C.cast(d, C.c_void_p), dict_size=4)
    assert not L.zxc_dstream_create(C.byref(do))
    for bs in (1000, 2048, 4 << 20):
        assert not L.zxc_cstream_create(C.byref(api._CompressOpts(level=3, block_size=bs)))
    for bs in (4096, 65536, 2 << 20):
        cs = L.zxc_cstream_create(C.byref(api._CompressOpts(level=9, block_size=bs)))  # (level clamped, not refused)
        assert cs
        L.zxc_cstream_free(cs)


@pytest.mark.parametrize("checksum", [False, True])
@pytest.mark.parametrize("out_chunk", [1, 5, 36, 4096])
def test_empty_stream_is_the_reference_s_36_bytes(ref, checksum, out_chunk):
    """header + EOF block + footer, drained through out buffers smaller than any of them (no device work involved)"""
    rc, arc = api.pstream_compress(b"", 4096, out_chunk, checksum=checksum)
    rrc, rarc = api.pstream_compress(b"", 4096, out_chunk, checksum=checksum, library=ref.lib)
    assert rc == rrc == 0 and arc == rarc and len(arc) == 36
    for in_chunk in (1, 7, 36, 100):
        assert api.pstream_decompress(arc, in_chunk, 64, checksum) == api.pstream_decompress(arc, in_chunk, 64, checksum, library=ref.lib) == (0, b"", 1, 36)


def test_end_then_any_call_is_rejected(L):
    """reference tests/test_pstream_api.c:595-645 and include/zxc_pstream.h:179-181"""
    cs = L.zxc_cstream_create(None)
    tiny = C.create_string_buffer(4)
    out = api._OutBuf(C.addressof(tiny), 4, 0)
    assert L.zxc_cstream_end(cs, C.byref(out)) == 12  # 16-byte header: 4 out, 12 pending
    src = C.create_string_buffer(b"x" * 64, 64)
    big = C.create_string_buffer(1024)
    assert L.zxc_cstream_compress(cs, C.byref(api._OutBuf(C.addressof(big), 1024, 0)), C.byref(api._InBuf(C.addressof(src), 64, 0))) == 0
    # (still in the header drain: _compress may go on; now finish)
    L.zxc_cstream_free(cs)
    cs = L.zxc_cstream_create(None)
    out = api._OutBuf(C.addressof(big), 1024, 0)
    assert L.zxc_cstream_end(cs, C.byref(out)) == 0 and out.pos == 36
    assert L.zxc_cstream_end(cs, C.byref(out)) == NULL_INPUT
    assert L.zxc_cstream_compress(cs, C.byref(out), C.byref(api._InBuf(None, 0, 0))) == NULL_INPUT
    L.zxc_cstream_free(cs)


def test_header_errors_are_the_reference_s_and_sticky(product, L, ref):
    junk = bytes(0xAA ^ i for i in range(16))
    assert api.pstream_decompress(junk, 16, 64)[0] == api.pstream_decompress(junk, 16, 64, library=ref.lib)[0] == BAD_MAGIC
    ds = L.zxc_dstream_create(None)
    src = C.create_string_buffer(junk, 16)
    buf = C.create_string_buffer(64)
    out = api._OutBuf(C.addressof(buf), 64, 0)
    assert L.zxc_dstream_decompress(ds, C.byref(out), C.byref(api._InBuf(C.addressof(src), 16, 0))) == BAD_MAGIC
    assert L.zxc_dstream_decompress(ds, C.byref(out), C.byref(api._InBuf(None, 0, 0))) == BAD_MAGIC
    assert L.zxc_dstream_finished(ds) == 0
    L.zxc_dstream_free(ds)
    _, empty = api.pstream_compress(b"", 64, 64, checksum=True)
    for pos in range(36):  # every byte of header / EOF block / footer flipped: same verdict as the reference, fed 1 and 36 bytes at a time
        bad = bytearray(empty)
        bad[pos] ^= 0x55
        for chunk in (1, 36):
            mine = api.pstream_decompress(bytes(bad), chunk, 64, True)
            theirs = api.pstream_decompress(bytes(bad), chunk, 64, True, library=ref.lib)
            if mine[0] == GPU_UNAVAILABLE and _no_gpu(product):
                continue  # (the flip turned the EOF block into a data block: that one needs the device, tests/test_gpu_pstream.py)
            assert mine == theirs, (pos, chunk, mine, theirs)
    for cut in range(36):  # truncated: not an error, just never finished
        mine = api.pstream_decompress(empty[:cut], 5, 64, True)
        assert mine == api.pstream_decompress(empty[:cut], 5, 64, True, library=ref.lib) == (0, b"", 0, cut)


def test_bytes_behind_the_footer_are_left_to_the_caller(ref):
    """include/zxc_pstream.h:241-244: DONE once the footer validates; trailing bytes are not consumed"""
    _, empty = api.pstream_compress(b"", 64, 64)
    for lib_ in (None, ref.lib):
        assert api.pstream_decompress(empty + b"trailing", 1 << 20, 64, library=lib_) == (0, b"", 1, 36)


def test_a_seek_table_behind_the_eof_block_is_skipped(ref):
    """an empty SEEKABLE archive of the reference (EOF block, SEK block, footer), fed in every chunking"""
    arc = ref.compress(b"", level=3, block_size=4096, seekable=True)
    for chunk in (1, 3, 8, 1000):
        mine = api.pstream_decompress(arc, chunk, 64)
        assert mine == api.pstream_decompress(arc, chunk, 64, library=ref.lib) and mine[2] == 1


def test_no_cpu_fallback_a_block_without_a_device_fails_loudly(product, L):
    if not _no_gpu(product):
        pytest.skip("a HIP device is present")
    rc, got = api.pstream_compress(b"abc" * 100, 4096, 4096)
    assert rc == GPU_UNAVAILABLE and len(got) == 16  # the header went out, the residual block could not be encoded
    cs = L.zxc_cstream_create(C.byref(api._CompressOpts(level=3, block_size=4096)))
    src = C.create_string_buffer(b"a" * 8192, 8192)
    buf = C.create_string_buffer(16384)
    out = api._OutBuf(C.addressof(buf), 16384, 0)
    inb = api._InBuf(C.addressof(src), 8192, 0)
    assert L.zxc_cstream_compress(cs, C.byref(out), C.byref(inb)) == GPU_UNAVAILABLE
    assert L.zxc_cstream_compress(cs, C.byref(out), C.byref(inb)) == GPU_UNAVAILABLE  # sticky
    assert L.zxc_cstream_end(cs, C.byref(out)) == GPU_UNAVAILABLE
    L.zxc_cstream_free(cs)


def test_no_cpu_fallback_on_the_decode_side(product, ref):
    if not _no_gpu(product):
        pytest.skip("a HIP device is present")
    arc = ref.compress(b"hello world, " * 1000, level=3, block_size=4096, seekable=False)
    rc, got, fin, used = api.pstream_decompress(arc, 1 << 20, 1 << 20)
    assert rc == GPU_UNAVAILABLE and got == b"" and fin == 0
