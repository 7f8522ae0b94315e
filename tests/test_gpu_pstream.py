"""Push streaming (include/zxc_pstream.h) on the GPU, against the UNMODIFIED reference driven through the same prototypes
(oracle/_ref/libzxc_ref.so): archives written by zxc_cstream_* are byte for byte this library's zxc_compress() output whatever
the chunking and decode with the reference's one-shot AND push decoders; zxc_dstream_* decodes the reference's archives (all
levels, seekable or not, with checksums) in any chunking to the reference's bytes; mutated archives give the reference's verdict,
code and delivered bytes; batches larger than one launch window, frames that straddle calls, irregular frames."""
import random

import pytest

import zxc_amd.api as api
from conftest import push_random_schedule

pytestmark = pytest.mark.gpu

CHUNKINGS = ((1 << 30, 1 << 30), (1 << 20, 1 << 20), (8192, 8192), (13 * 1024, 7 * 1024), (511, 7000), (137, 53))


@pytest.fixture(scope="module")
def gpu(product):
    assert product.lib().zxc_mi355x_device_count() >= 1, "no HIP device"
    product.lib().zxc_mi355x_set_device(0)
    return product


def _text(rng, n, vocab=60):
    words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz ,.") for _ in range(rng.randrange(3, 10))) for _ in range(vocab)]
    out = bytearray()
    while len(out) < n:
        out += rng.choice(words)
    return bytes(out[:n])


def _mixed(rng, n):
    """text, a run, random bytes, text: GLO / RLE-literal / RAW blocks in one stream"""
    q = n // 4
    return _text(rng, q) + bytes([7]) * q + rng.randbytes(q) + _text(rng, n - 3 * q, vocab=400)


@pytest.mark.parametrize("block_size", [4096, 65536, 512 * 1024])
@pytest.mark.parametrize("checksum", [False, True])
def test_cstream_writes_zxc_compress_s_bytes_in_any_chunking(gpu, ref, block_size, checksum):
    rng = random.Random(block_size + checksum)
    data = _mixed(rng, 5 * block_size + block_size // 3 + 17)
    want = gpu.compress(data, level=3, block_size=block_size, seekable=False, checksum=checksum)
    for in_chunk, out_chunk in CHUNKINGS:
        rc, arc = api.pstream_compress(data, in_chunk, out_chunk, level=3, block_size=block_size, checksum=checksum)
        assert rc == 0 and arc == want, (in_chunk, out_chunk, rc, len(arc), len(want))
    # the reference reads it: one-shot and push
    rc, got = ref.decompress(want, len(data), checksum=checksum)
    assert rc == len(data) and got == data
    assert api.pstream_decompress(want, 4000, 9000, checksum, library=ref.lib) == (0, data, 1, len(want))


@pytest.mark.parametrize("level", [1, 2, 4, 5, 6, 7])
def test_cstream_levels_round_trip_through_the_reference(gpu, ref, level):
    rng = random.Random(level)
    data = _mixed(rng, 70 * 1024)
    rc, arc = api.pstream_compress(data, 16 * 1024, 16 * 1024, level=level, block_size=65536, checksum=True)
    assert rc == 0 and arc == gpu.compress(data, level=level, block_size=65536, seekable=False, checksum=True)
    assert api.pstream_decompress(arc, 16 * 1024, 16 * 1024, True, library=ref.lib) == (0, data, 1, len(arc))
    assert api.pstream_decompress(arc, 16 * 1024, 16 * 1024, True) == (0, data, 1, len(arc))


def test_cstream_one_byte_at_a_time_and_exact_block_multiples(gpu, ref):
    rng = random.Random(5)
    for n in (1, 4095, 4096, 4097, 8192, 3 * 4096):
        data = _text(rng, n)
        want = gpu.compress(data, level=3, block_size=4096, seekable=False, checksum=True)
        for in_chunk, out_chunk in ((1, 4096), (n, 1), (4096, 37)):
            assert api.pstream_compress(data, in_chunk, out_chunk, level=3, block_size=4096, checksum=True) == (0, want), (n, in_chunk)
        assert ref.decompress(want, n, checksum=True) == (n, data)


@pytest.mark.parametrize("level", [1, 3, 5, 6, 7])
@pytest.mark.parametrize("seekable", [False, True])
def test_dstream_reads_the_reference_s_archives_in_any_chunking(gpu, ref, level, seekable):
    rng = random.Random(10 * level + seekable)
    bs = 16384
    data = _mixed(rng, 9 * bs + 1234)
    for checksum in (False, True):
        arc = ref.compress(data, level, bs, seekable, checksum)
        for in_chunk, out_chunk in CHUNKINGS:
            for verify in ((False, True) if checksum else (False,)):
                got = api.pstream_decompress(arc, in_chunk, out_chunk, verify)
                assert got == (0, data, 1, len(arc)), (checksum, in_chunk, out_chunk, verify, got[0], len(got[1]), got[2:])
    small = ref.compress(data[:8192], level, 4096, seekable, True)
    assert api.pstream_decompress(small, 1, 4096, True) == (0, data[:8192], 1, len(small))  # the 1-byte feeder


def test_batches_larger_than_one_launch_window(gpu, ref, monkeypatch):
    """72 MiB through 32 MiB windows (default: 128): one call's input holds more blocks than a launch takes (both directions), with
    out buffers larger and smaller than a window's output"""
    monkeypatch.setenv("ZXC_MI355X_PSTREAM_WINDOW_MIB", "32")
    rng = random.Random(77)
    unit = _mixed(rng, 3 << 20)
    data = b"".join(unit[i:] + unit[:i] for i in range(0, 24 * 4099, 4099))  # 24 rotations: 72 MiB
    bs = 65536
    want = gpu.compress(data, level=3, block_size=bs, seekable=False, checksum=True)
    for in_chunk, out_chunk in ((1 << 30, 1 << 30), (40 << 20, 1 << 20), (5 << 20, 50 << 20)):
        rc, arc = api.pstream_compress(data, in_chunk, out_chunk, level=3, block_size=bs, checksum=True)
        assert rc == 0 and arc == want, (in_chunk, out_chunk)
    rarc = ref.compress(data, 3, bs, True, True)
    for in_chunk, out_chunk in ((1 << 30, 1 << 30), (40 << 20, 1 << 20), (5 << 20, 50 << 20), (3 << 20, 3 << 20)):
        got = api.pstream_decompress(rarc, in_chunk, out_chunk, True)
        assert got[0] == 0 and got[2:] == (1, len(rarc)) and got[1] == data, (in_chunk, out_chunk, got[0], len(got[1]))


def _blocks(arc):
    out, ip = [], 16
    while arc[ip] != 255:
        csz = int.from_bytes(arc[ip + 3:ip + 7], "little")
        out.append((ip, 8 + csz))
        ip += 8 + csz
    return out, ip


def test_irregular_frames_short_blocks_in_the_middle(gpu, ref):
    """a's blocks (the last one short) followed by b's in ONE frame: legal, never written by an encoder; the reference's push
    decoder appends what each block yields"""
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
            assert api.pstream_decompress(fr, in_chunk, out_chunk) == want, (k, in_chunk, out_chunk)


def test_mutants_get_the_reference_s_verdict_code_and_delivered_bytes(gpu, ref):
    """400 mutants (bit flips, stomps, cuts, flipped framing bytes) of four archives, each fed in a random chunking to both push
    decoders: same return code, same bytes delivered before it, same finished flag. (Input consumed is compared on success only:
    behind a failing block this decoder has taken the rest of its batch.)"""
    rng = random.Random(2026)
    bs = 4096
    data = _mixed(rng, 24 * bs + 99)
    arcs = [ref.compress(data, lv, bs, sk, ck) for lv, sk, ck in ((3, False, False), (3, True, False), (6, False, True), (1, True, True), (3, False, True))]
    failed = ok = 0
    for it in range(400):
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
        else:  # a block header's type / size / crc byte, or the tail
            blocks, eofb = _blocks(arcs[0])  # (arcs[0]: no trailers, no seek table)
            m = bytearray(arcs[0])
            o = rng.choice(blocks)[0] if rng.random() < 0.7 else eofb
            m[o + rng.randrange(8 if o != eofb else 20)] ^= 1 << rng.randrange(8)
        m = bytes(m)
        in_chunk, out_chunk = rng.choice(CHUNKINGS[1:])
        verify = rng.random() < 0.5
        want = api.pstream_decompress(m, in_chunk, out_chunk, verify, library=ref.lib)
        got = api.pstream_decompress(m, in_chunk, out_chunk, verify)
        assert got[:3] == want[:3], (it, kind, in_chunk, out_chunk, verify, got[0], want[0], len(got[1]), len(want[1]), got[2], want[2])
        if want[0] == 0:
            assert got[3] == want[3]
        failed += want[0] < 0
        ok += want[0] == 0 and want[2] == 1
    assert failed >= 100 and ok >= 20, (failed, ok)


def test_framing_bytes_of_an_empty_stream_flipped(gpu, ref):
    """every byte of header / EOF block / footer flipped (tests/test_pstream_cpu.py skips the flips that make a data block)"""
    _, empty = api.pstream_compress(b"", 64, 64, checksum=True)
    for pos in range(36):
        bad = bytearray(empty)
        bad[pos] ^= 0x55
        for chunk in (1, 36):
            got = api.pstream_decompress(bytes(bad), chunk, 64, True)
            want = api.pstream_decompress(bytes(bad), chunk, 64, True, library=ref.lib)
            # (input consumed is compared unless a BLOCK failed: behind it this decoder has looked at the rest of its batch)
            assert got[:3] == want[:3] and (got[3] == want[3] or (want[0] < 0 and pos in range(16, 24))), (pos, chunk, got, want)


def test_random_call_schedules(gpu, ref):
    """every call with another in / out size (zero included): the archive is zxc_compress's, the reference reads it, the decoder
    returns the source from a seekable archive of the reference and leaves the bytes behind the footer alone"""
    bs = 4096
    for seed in range(8):
        rng = random.Random(200 + seed)
        data = _mixed(rng, rng.randrange(1, 40 * bs))
        checksum = bool(seed & 1)
        want = gpu.compress(data, level=3, block_size=bs, seekable=False, checksum=checksum)
        arc = ref.compress(data, 3, bs, True, checksum)
        blob, dec, used = push_random_schedule(gpu.lib(), data, arc + b"trailing", random.Random(seed), bs, checksum)
        assert blob == want and dec == data and used == len(arc), (seed, len(blob), len(want), len(dec), used)
        assert ref.decompress(blob, len(data), checksum=checksum) == (len(data), data)
