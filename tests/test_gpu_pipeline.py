"""GPU tests of the host API's pipelines (run with -m gpu on an MI355X): zxc_decompress and the seekable range calls move a
frame through the device in pieces — upload of piece i+1 beside the decode of piece i beside the download of piece i-1 — and
every intricate corner of that (multi-piece frames, a failing block in a later piece, an irregular frame across pieces,
concurrent callers, ranges that straddle pieces) must give the UNMODIFIED reference's bytes and codes. ZXC_MI355X_FRAME_BATCH_MIB=1
shrinks the pieces so that small inputs cross many of them. Plus the differential fuzz (tools/fuzzdiff.py, 2 000 mutants) and a
subset of the parity sweep (tools/sweep.py) that used to run by hand only."""
import os
import random
import subprocess
import sys
import threading

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(product):
    assert product.lib().zxc_mi355x_device_count() >= 1, "no HIP device"
    product.lib().zxc_mi355x_set_device(0)
    return product


def _text(rng, n):
    words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz ,.") for _ in range(rng.randrange(3, 10))) for _ in range(60)]
    out = bytearray()
    while len(out) < n:
        out += rng.choice(words)
    return bytes(out[:n])


def _blocks(arc):
    """[(offset, physical size)] of a non-seekable, checksum-free archive's blocks and the offset of its EOF block"""
    out, ip = [], 16
    while arc[ip] != 255:
        csz = int.from_bytes(arc[ip + 3:ip + 7], "little")
        out.append((ip, 8 + csz))
        ip += 8 + csz
    return out, ip


def _splice(a, b):
    """a's blocks (its last one decodes short) followed by b's blocks in ONE frame: legal, never written by the reference encoder"""
    A, _ = _blocks(a)
    B, eofb = _blocks(b)
    tot = int.from_bytes(a[-12:-4], "little") + int.from_bytes(b[-12:-4], "little")
    return (a[:16] + b"".join(a[o:o + n] for o, n in A) + b"".join(b[o:o + n] for o, n in B) + b[eofb:eofb + 8] +
            tot.to_bytes(8, "little") + (0).to_bytes(4, "little"))


def test_multi_piece_frames_round_trip(gpu, ref, monkeypatch):
    """A frame of many pieces (1 MiB pieces: 256 blocks of 4 KiB, 16 of 64 KiB) decodes to the reference's bytes; with and without
    per-block checksums; sizes around the piece borders."""
    monkeypatch.setenv("ZXC_MI355X_FRAME_BATCH_MIB", "1")
    rng = random.Random(21)
    data = _text(rng, 7 * (1 << 20) + 12345)
    for bs, ck in ((4096, False), (65536, True), (65536, False)):
        for n in (len(data), 1 << 20, (1 << 20) + 1, (2 << 20) - 1, 3 * (1 << 20) + bs):
            comp = ref.compress(data[:n], 3, bs, False, ck)
            assert gpu.decompress(comp, checksum=ck) == data[:n], (bs, ck, n)


def test_failing_block_in_a_later_piece_reports_the_references_code(gpu, ref, monkeypatch):
    """First failing block in stream order wins, whatever piece it is in and whatever runs ahead of it (reference: the sequential loop
    of zxc_decompress_frame, src/lib/zxc_dispatch.c:912-1001): mutants of blocks 300 / 600 / both of an 800-block frame."""
    monkeypatch.setenv("ZXC_MI355X_FRAME_BATCH_MIB", "1")
    rng = random.Random(22)
    bs = 4096
    data = _text(rng, 800 * bs)
    comp = ref.compress(data, 3, bs, False, False)
    blocks, _ = _blocks(comp)
    checked = failed = 0
    for trial in range(24):
        m = bytearray(comp)
        for bi in ((300,), (600,), (300, 600), (600, 300, 799))[trial % 4]:
            o, n = blocks[bi]
            kind = rng.randrange(3)
            if kind == 0:  # header fields of the block: n_seq / n_lit / codings
                m[o + 8 + rng.randrange(12)] ^= 1 << rng.randrange(8)
            elif kind == 1:  # a byte anywhere in the payload
                m[o + 8 + rng.randrange(n - 8)] = rng.randrange(256)
            else:  # offsets / extras region
                p = o + n - 1 - rng.randrange(min(64, n - 9))
                m[p] = rng.choice((0, 0xFF, 0xE0, 0x80))
        m = bytes(m)
        want_rc, want = ref.decompress(m, len(data))
        rc, got = gpu.decompress(m, len(data), raise_on_error=False)
        assert rc == want_rc, (trial, rc, want_rc)
        if rc >= 0:
            assert got == want
        failed += rc < 0
        checked += 1
    assert checked == 24 and failed >= 6


def test_irregular_frame_across_pieces(gpu, ref, monkeypatch):
    """A non-final block that decodes short of block_size (legal; the reference decoder just appends what a block yields): in the
    first piece and in the second one, with prefetched pieces behind it."""
    monkeypatch.setenv("ZXC_MI355X_FRAME_BATCH_MIB", "1")
    rng = random.Random(23)
    bs = 4096
    for k in (5, 300, 511):
        dA, dB = _text(rng, k * bs + 1000), _text(rng, 600 * bs + 77)
        fr = _splice(ref.compress(dA, 3, bs, False, False), ref.compress(dB, 3, bs, False, False))
        rc, want = ref.decompress(fr, len(dA) + len(dB))
        assert rc == len(dA) + len(dB) and want == dA + dB
        assert gpu.decompress(fr) == dA + dB, k
        # and a capacity one byte short of what the frame yields: the reference's code
        rc_ref, _ = ref.decompress(fr, len(dA) + len(dB) - 1)
        rc, _ = gpu.decompress(fr, len(dA) + len(dB) - 1, raise_on_error=False)
        assert rc == rc_ref < 0


def test_concurrent_callers_on_one_device(gpu, ref, monkeypatch):
    """Four threads in zxc_decompress / zxc_seekable_decompress_range at once (they share the device's staging arenas and stream
    slots; nobody waits for a second arena while holding one): every call returns its own frame's bytes."""
    monkeypatch.setenv("ZXC_MI355X_FRAME_BATCH_MIB", "1")
    rng = random.Random(24)
    frames = []
    for i in range(4):
        d = _text(rng, (3 + i) * (1 << 20) + i * 1001)
        frames.append((d, ref.compress(d, 3, 65536, True, False)))
    errs = []

    def work(i):
        try:
            d, c = frames[i]
            gpu.lib().zxc_mi355x_set_device(0)
            for _ in range(3):
                assert gpu.decompress(c) == d
                s = gpu.Seekable(c)
                a = 70000 * (i + 1)
                assert s.decompress_range(a, len(d) - a - 5) == d[a:len(d) - 5]
                s.close()
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


def test_seekable_ranges_across_pieces(gpu, ref, monkeypatch):
    """The range call moves its blocks through the device in pieces too: ranges that start / end inside blocks, inside pieces, at
    piece borders; a corrupt block inside the range (the reference's code) and outside it (never looked at)."""
    monkeypatch.setenv("ZXC_MI355X_FRAME_BATCH_MIB", "1")
    rng = random.Random(25)
    bs = 65536
    data = _text(rng, 9 * (1 << 20) + 4321)
    comp = ref.compress(data, 3, bs, True, False)
    s = gpu.Seekable(comp)
    cases = [(0, len(data)), (1, len(data) - 1), ((1 << 20) - 1, 2), (1 << 20, 1 << 20), ((1 << 20) - 70000, (3 << 20) + 140001),
             (len(data) - 1, 1), (5 * (1 << 20) + 17, 1)]
    for _ in range(8):
        a = rng.randrange(len(data))
        cases.append((a, rng.randrange(1, len(data) - a + 1)))
    for a, n in cases:
        assert s.decompress_range(a, n) == data[a:a + n], (a, n)
    jobs = s.plan()
    s.close()
    bad = bytearray(comp)
    o = int(jobs["comp_off"][40]) + 8  # block 40's n_seq field: far too many sequences for its payload
    bad[o:o + 4] = (0x00FFFFFF).to_bytes(4, "little")
    u = gpu.Seekable(bytes(bad))
    want_rc, _ = ref.seekable_range_mt(bytes(bad), 0, len(data), 1)
    rc, _ = u.decompress_range(0, len(data), raise_on_error=False)
    assert rc == want_rc < 0
    assert u.decompress_range(41 * bs, 3 * (1 << 20)) == data[41 * bs:41 * bs + 3 * (1 << 20)]  # the damaged block is outside
    assert u.decompress_range(0, 40 * bs) == data[:40 * bs]
    u.close()


def test_differential_fuzz_2000_mutants():
    """tools/fuzzdiff.py: 2 000 mutated archives (bit flips, byte stomps, truncations), device vs oracle: same accept / reject, same
    code, same bytes."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzzdiff.py"), "2000"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("2000 mutants") and last.endswith(" 0 mismatches"), r.stdout[-2000:]


def test_parity_sweep_subset(ref):
    """tools/sweep.py on 1 MiB per class (SWEEP_SUBSET=1: three data classes): reference encoder -> device decoder (buffer and seekable
    API), device encoder -> reference decoder, levels 1-7 x block sizes 4 KiB .. 2 MiB."""
    env = dict(os.environ, SWEEP_SUBSET="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep.py"), "1"], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert " 0 failures" in last and int(last.split()[0]) >= 150, r.stdout[-2000:]
