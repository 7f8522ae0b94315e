"""The HOST side of the library without a GPU: zxc_amd/csrc/zxc_host.c (and what it includes) compiled over tests/mock_device — a
"device" of host memory whose two kernels are the UNMODIFIED reference's Block API. What runs here is the product's own framing,
batching, error precedence and state machines (push streaming, the piece pipeline of zxc_decompress / zxc_compress / the
seekable ranges); since the block codec underneath is the reference's, archives must be BYTE FOR BYTE the reference's and every
verdict the reference's. The same properties on the real device: tests/test_gpu_pstream.py, tests/test_gpu_pipeline.py."""
import random

import pytest

import zxc_amd.api as api
from conftest import push_random_schedule

CHUNKINGS = ((1 << 30, 1 << 30), (8192, 8192), (13 * 1024, 700), (511, 7000), (137, 53), (1, 4096))


def _text(rng, n, vocab=60):
    words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz ,.") for _ in range(rng.randrange(3, 10))) for _ in range(vocab)]
    out = bytearray()
    while len(out) < n:
        out += rng.choice(words)
    return bytes(out[:n])


def _mixed(rng, n):
    q = n // 4
    return _text(rng, q) + bytes([7]) * q + rng.randbytes(q) + _text(rng, n - 3 * q, vocab=400)


def _blocks(arc):
    out, ip = [], 16
    while arc[ip] != 255:
        csz = int.from_bytes(arc[ip + 3:ip + 7], "little")
        out.append((ip, 8 + csz))
        ip += 8 + csz
    return out, ip


def _as_ref(mockdev):
    import oracle_py
    r = oracle_py.Ref.__new__(oracle_py.Ref)
    r.lib = oracle_py.bind_zxc_api(mockdev)
    return r


@pytest.mark.parametrize("checksum", [False, True])
def test_push_compress_writes_the_reference_s_archive_in_any_chunking(mockdev, ref, checksum):
    rng = random.Random(3 + checksum)
    bs = 4096
    for n in (0, 1, bs - 1, bs, bs + 1, 20 * bs + 77):
        data = _mixed(rng, n)
        want = ref.compress(data, 3, bs, False, checksum)
        assert api.pstream_compress(data, 8192, 8192, level=3, block_size=bs, checksum=checksum, library=ref.lib) == (0, want)
        for in_chunk, out_chunk in CHUNKINGS:
            got = api.pstream_compress(data, in_chunk, out_chunk, level=3, block_size=bs, checksum=checksum, library=mockdev)
            assert got == (0, want), (n, in_chunk, out_chunk, got[0], len(got[1]), len(want))


@pytest.mark.parametrize("seekable", [False, True])
def test_push_decompress_in_any_chunking(mockdev, ref, seekable):
    rng = random.Random(5 + seekable)
    bs = 4096
    data = _mixed(rng, 20 * bs + 77)
    for level, checksum in ((3, False), (3, True), (6, True)):
        arc = ref.compress(data, level, bs, seekable, checksum)
        for in_chunk, out_chunk in CHUNKINGS:
            for verify in (False, True):
                got = api.pstream_decompress(arc, in_chunk, out_chunk, verify, library=mockdev)
                assert got == (0, data, 1, len(arc)), (level, checksum, in_chunk, out_chunk, verify, got[0], len(got[1]), got[2:])


def test_push_streams_across_launch_windows(mockdev, ref, monkeypatch):
    """40 MiB of 4 KiB blocks = 10 240 blocks through windows of 8 192 (32 MiB instead of the default 128): calls whose input holds
    more than one launch"""
    monkeypatch.setenv("ZXC_MI355X_PSTREAM_WINDOW_MIB", "32")
    rng = random.Random(7)
    unit = _mixed(rng, 1 << 20)
    data = b"".join(unit[i:] + unit[:i] for i in range(0, 40 * 4099, 4099))
    bs = 4096
    want = ref.compress(data, 1, bs, False, True)
    for in_chunk, out_chunk in ((1 << 30, 1 << 30), (36 << 20, 1 << 20), (3 << 20, 38 << 20)):
        assert api.pstream_compress(data, in_chunk, out_chunk, level=1, block_size=bs, checksum=True, library=mockdev) == (0, want)
        got = api.pstream_decompress(want, in_chunk, out_chunk, True, library=mockdev)
        assert got[0] == 0 and got[2:] == (1, len(want)) and got[1] == data, (in_chunk, out_chunk)


def test_push_decompress_irregular_frames(mockdev, ref):
    """a's blocks (the last one short) then b's in one frame: a second launch with capacity-sized slots"""
    rng = random.Random(23)
    bs = 4096
    for k in (1, 5, 40):
        dA, dB = _text(rng, k * bs + 1000), _text(rng, 30 * bs + 77)
        a, b = ref.compress(dA, 3, bs, False, False), ref.compress(dB, 3, bs, False, False)
        A, _ = _blocks(a)
        B, eofb = _blocks(b)
        fr = (a[:16] + b"".join(a[o:o + n] for o, n in A) + b"".join(b[o:o + n] for o, n in B) + b[eofb:eofb + 8] +
              (len(dA) + len(dB)).to_bytes(8, "little") + bytes(4))
        for in_chunk, out_chunk in ((1 << 20, 1 << 20), (3000, 1 << 20), (1 << 20, 5000), (777, 333)):
            want = api.pstream_decompress(fr, in_chunk, out_chunk, library=ref.lib)
            assert want == (0, dA + dB, 1, len(fr))
            assert api.pstream_decompress(fr, in_chunk, out_chunk, library=mockdev) == want, (k, in_chunk, out_chunk)


def test_push_decompress_mutants_get_the_reference_s_verdict(mockdev, ref):
    """600 mutants, random chunkings: same code, same bytes delivered in front of it, same finished flag; the same input consumed
    whenever the stream did not fail"""
    rng = random.Random(2026)
    bs = 4096
    data = _mixed(rng, 24 * bs + 99)
    arcs = [ref.compress(data, lv, bs, sk, ck) for lv, sk, ck in ((3, False, False), (3, True, False), (6, False, True), (1, True, True), (3, False, True))]
    failed = ok = 0
    for it in range(600):
        m = bytearray(rng.choice(arcs))
        kind = rng.randrange(5)
        if kind == 0:
            for _ in range(rng.choice((1, 1, 2, 4))):
                m[rng.randrange(16, len(m))] ^= 1 << rng.randrange(8)
        elif kind == 1:
            p = rng.randrange(16, len(m))
            m[p:p + rng.randrange(1, 9)] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
        elif kind == 2:
            m[rng.randrange(16, len(m) - 8)] = rng.choice((0, 0xFF, 0x80, 0x7F, 0xE0))
        elif kind == 3:
            cut = rng.randrange(20, len(m))
            del m[cut:cut + rng.randrange(1, 64)]
        else:  # a framing byte: a block header's type / size / crc, the EOF block, the footer
            blocks, eofb = _blocks(arcs[0])
            m = bytearray(arcs[0])
            o = rng.choice(blocks)[0] if rng.random() < 0.7 else eofb
            m[o + rng.randrange(8 if o != eofb else 20)] ^= 1 << rng.randrange(8)
        m = bytes(m)
        in_chunk, out_chunk = rng.choice(CHUNKINGS[1:5])
        verify = rng.random() < 0.5
        want = api.pstream_decompress(m, in_chunk, out_chunk, verify, library=ref.lib)
        got = api.pstream_decompress(m, in_chunk, out_chunk, verify, library=mockdev)
        assert got[:3] == want[:3], (it, kind, in_chunk, out_chunk, verify, got[0], want[0], len(got[1]), len(want[1]))
        if want[0] == 0:
            assert got[3] == want[3]
        failed += want[0] < 0
        ok += want[0] == 0 and want[2] == 1
    assert failed >= 200 and ok >= 50, (failed, ok)


def test_piece_pipeline_frames_and_ranges(mockdev, ref, monkeypatch):
    """zxc_compress / zxc_decompress / zxc_seekable_decompress_range through the piece pipeline (1 MiB pieces, two producer threads,
    the in-order sink) on the mock device: the reference's archive bytes, the source back, the reference's code for a failing block
    in a later piece"""
    monkeypatch.setenv("ZXC_MI355X_FRAME_BATCH_MIB", "1")
    m = _as_ref(mockdev)
    rng = random.Random(31)
    bs = 4096
    data = _mixed(rng, 5 * (1 << 20) + 12345)
    for seekable, checksum in ((False, False), (True, True)):
        want = ref.compress(data, 3, bs, seekable, checksum)
        assert m.compress(data, 3, bs, seekable, checksum) == want
        rc, got = m.decompress(want, len(data), checksum=checksum)
        assert rc == len(data) and got == data
    arc = ref.compress(data, 3, bs, True, False)
    for _ in range(12):
        off = rng.randrange(len(data))
        ln = rng.randrange(1, min(len(data) - off, 2 << 20) + 1)
        rc, got = m.seekable_range_mt(arc, off, ln, rng.choice((1, 4)))
        assert rc == ln and got == data[off:off + ln]
    blocks, _ = _blocks(ref.compress(data, 3, bs, False, False))
    plain = ref.compress(data, 3, bs, False, False)
    for bi in (10, 400, 1200):
        bad = bytearray(plain)
        o, n = blocks[bi]
        bad[o + 8 + rng.randrange(12)] ^= 0x40
        want_rc, _ = ref.decompress(bytes(bad), len(data))
        rc, _ = m.decompress(bytes(bad), len(data))
        assert rc == want_rc, (bi, rc, want_rc)


def test_push_decompress_many_pieces_irregular_and_failing_blocks(mockdev, ref, monkeypatch):
    """1 MiB pieces (256 blocks of 4 KiB): a short block in the first piece / in a later one / at a piece border, and a failing
    block in a later piece, with whole-input calls (everything lands in out) and with windows staged through a small out"""
    monkeypatch.setenv("ZXC_MI355X_FRAME_BATCH_MIB", "1")
    monkeypatch.setenv("ZXC_MI355X_PSTREAM_WINDOW_MIB", "2")
    rng = random.Random(41)
    bs = 4096
    for k in (5, 300, 511):
        dA, dB = _text(rng, k * bs + 1000), _text(rng, 600 * bs + 77)
        a, b = ref.compress(dA, 3, bs, False, False), ref.compress(dB, 3, bs, False, False)
        A, _ = _blocks(a)
        B, eofb = _blocks(b)
        fr = (a[:16] + b"".join(a[o:o + n] for o, n in A) + b"".join(b[o:o + n] for o, n in B) + b[eofb:eofb + 8] +
              (len(dA) + len(dB)).to_bytes(8, "little") + bytes(4))
        for in_chunk, out_chunk in ((1 << 30, 1 << 30), (1 << 30, 100000), (700001, 1 << 30), (300000, 50000)):
            assert api.pstream_decompress(fr, in_chunk, out_chunk, library=mockdev) == (0, dA + dB, 1, len(fr)), (k, in_chunk, out_chunk)
    data = _text(rng, 1500 * bs)
    arc = ref.compress(data, 3, bs, False, True)
    blocks = []
    ip = 16
    while arc[ip] != 255:
        csz = int.from_bytes(arc[ip + 3:ip + 7], "little")
        blocks.append((ip, 8 + csz + 4))
        ip += 8 + csz + 4
    for bi in ((3,), (700,), (1499,), (300, 900)):
        bad = bytearray(arc)
        for x in bi:
            o, n = blocks[x]
            bad[o + 8 + rng.randrange(n - 12)] ^= 0x10
        for in_chunk, out_chunk in ((1 << 30, 1 << 30), (1 << 30, 100000), (300000, 1 << 30)):
            for verify in (False, True):
                want = api.pstream_decompress(bytes(bad), in_chunk, out_chunk, verify, library=ref.lib)
                got = api.pstream_decompress(bytes(bad), in_chunk, out_chunk, verify, library=mockdev)
                assert got[:3] == want[:3], (bi, in_chunk, out_chunk, verify, got[0], want[0], len(got[1]), len(want[1]))


def test_file_stream_engine(mockdev, ref, tmp_path, monkeypatch):
    """zxc_stream_compress / zxc_stream_decompress (the FILE* callers: reader, batches, in-order writer thread) on the mock device:
    the archive file is byte for byte the reference's zxc_compress of the same bytes, both FILE* decoders give the source back,
    and a mutated archive gets the reference FILE* decoder's verdict (an error or the same bytes)."""
    import hashlib
    monkeypatch.setenv("ZXC_STREAM_BATCH_BYTES", str(2 << 20))  # 512 KiB of source per launch: several batches
    rng = random.Random(51)
    data = _mixed(rng, 3 * (1 << 20) + 4321)
    src = tmp_path / "in.bin"
    src.write_bytes(data)
    arc, back = tmp_path / "a.zxc", tmp_path / "back.bin"
    for level, bs, seekable, ck in ((1, 65536, False, False), (3, 4096, True, True), (5, 512 * 1024, True, False)):
        n = api.stream_compress(str(src), str(arc), level=level, block_size=bs, seekable=seekable, checksum=ck, library=mockdev)
        assert n == arc.stat().st_size
        assert arc.read_bytes() == ref.compress(data, level, bs, seekable, ck), (level, bs, seekable, ck)
        for lib_ in (mockdev, ref.lib):
            assert api.stream_decompress(str(arc), str(back), checksum=ck, library=lib_) == len(data)
            assert hashlib.sha256(back.read_bytes()).digest() == hashlib.sha256(data).digest()
        assert api.stream_decompress(str(arc), None, checksum=ck, library=mockdev) == len(data)  # integrity only
        assert api.stream_get_decompressed_size(str(arc), library=mockdev) == len(data)
    comp = ref.compress(data, 3, 4096, False, True)
    bad = tmp_path / "bad.zxc"
    for _ in range(40):
        m = bytearray(comp)
        kind = rng.randrange(3)
        if kind == 0:
            m[rng.randrange(16, len(m))] ^= 1 << rng.randrange(8)
        elif kind == 1:
            del m[rng.randrange(16, len(m)):]
        else:
            p = rng.randrange(16, len(m) - 8)
            m[p:p + 4] = rng.randbytes(4)
        bad.write_bytes(bytes(m))
        want = api.stream_decompress(str(bad), str(tmp_path / "w.bin"), checksum=True, library=ref.lib)
        got = api.stream_decompress(str(bad), str(back), checksum=True, library=mockdev)
        assert (got < 0) == (want < 0), (kind, got, want)
        if want >= 0:
            assert got == want and back.read_bytes() == (tmp_path / "w.bin").read_bytes()
    empty = tmp_path / "empty.bin"
    empty.write_bytes(b"")
    assert api.stream_compress(str(empty), str(arc), library=mockdev) == arc.stat().st_size == 36
    assert arc.read_bytes() == ref.compress(b"", 3, 65536, True, False)


def test_push_streams_from_four_threads_at_once(mockdev, ref):
    """one context per thread, all of them through the same staging arenas and pipeline (nobody waits for a second arena while
    holding one): every thread gets its own bytes"""
    import threading
    rng = random.Random(61)
    bs = 4096
    datas = [_mixed(rng, (300 + 50 * i) * bs + i) for i in range(4)]
    wants = [ref.compress(d, 3, bs, False, True) for d in datas]
    errs = []

    def work(i):
        try:
            for _ in range(3):
                assert api.pstream_compress(datas[i], 1 << 30, 8 << 20, level=3, block_size=bs, checksum=True, library=mockdev) == (0, wants[i])
                assert api.pstream_decompress(wants[i], 200000, 8 << 20, True, library=mockdev) == (0, datas[i], 1, len(wants[i]))
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)[:300]))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_push_streams_under_random_call_schedules(mockdev, ref):
    """every call with another in / out size (zero included): the archive is still the reference's zxc_compress bytes, the decoder
    still returns the source, consumes the whole archive and nothing behind it; the reference's own push API under the same schedule
    agrees"""
    bs = 4096
    for seed in range(12):
        rng = random.Random(100 + seed)
        data = _mixed(rng, rng.randrange(1, 40 * bs))
        checksum = bool(seed & 1)
        want = ref.compress(data, 3, bs, False, checksum)
        seekable_arc = ref.compress(data, 3, bs, True, checksum)
        for lib_ in (mockdev, ref.lib):
            blob, dec, used = push_random_schedule(lib_, data, seekable_arc + b"trailing", random.Random(seed), bs, checksum)
            assert blob == want and dec == data and used == len(seekable_arc), (seed, lib_ is mockdev, len(blob), len(want), len(dec), used)


def test_static_cctx_with_a_dictionary_keeps_its_lock(mockdev):
    """zxc_compress_block on a static (caller-workspace) context checked the locked size against the effective
    [dict | block] size and then stored the base size — the second identical call failed with ZXC_ERROR_BAD_BLOCK_SIZE (-14).
    Reference: src/lib/zxc_dispatch.c:1653-1667."""
    import ctypes as C
    L = mockdev
    L.zxc_static_cctx_workspace_size.restype = C.c_size_t
    L.zxc_static_cctx_workspace_size.argtypes = [C.c_size_t, C.c_int]
    L.zxc_init_static_cctx.restype = C.c_void_p
    L.zxc_init_static_cctx.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.zxc_compress_block.restype = C.c_int64
    L.zxc_compress_block.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    ws_size = L.zxc_static_cctx_workspace_size(65536, 3)
    assert ws_size > 0
    ws = C.create_string_buffer(ws_size + 64)
    base = (C.addressof(ws) + 63) & ~63
    init = api._CompressOpts(level=3, block_size=65536)
    ctx = L.zxc_init_static_cctx(base, ws_size, C.byref(init))
    assert ctx
    rng = random.Random(3)
    dict_ = _text(rng, 32768)
    dbuf = C.create_string_buffer(dict_, len(dict_))
    src = _text(rng, 3000)
    o = api._CompressOpts(level=3, block_size=4096, dict=C.addressof(dbuf), dict_size=len(dict_))  # [32 KiB | 4 KiB] -> effective 64 KiB
    dst = C.create_string_buffer(8192)
    sizes = [L.zxc_compress_block(ctx, src, len(src), dst, len(dst), C.byref(o)) for _ in range(3)]
    assert sizes[0] > 0 and sizes == [sizes[0]] * 3, sizes
