import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Tests that hand torch streams / device tensors to the library need ONE HIP runtime in the process: torch
# ships its own libamdhip64, and whichever copy is loaded first is the one both sides must share. Loading
# torch before libzxc_mi355x.so (as bench.py does) makes that torch's.
try:
    import torch  # noqa: F401
except Exception:  # torch is plumbing for a few tests only
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    return json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    if not os.path.exists(oracle_py.ORACLE_SO):
        oracle_py.build()
    return oracle_py.Oracle()


@pytest.fixture(scope="session")
def ref():
    import oracle_py
    if not oracle_py.Ref.available():
        pytest.skip("oracle/_ref/libzxc_ref.so not built (needs /root/reference at build time)")
    return oracle_py.Ref()


@pytest.fixture(scope="session")
def synth_inputs():
    import make_golden
    return make_golden.synth_inputs()


@pytest.fixture(scope="session")
def product():
    import zxc_amd
    if not os.path.exists(zxc_amd.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    zxc_amd.lib()
    return zxc_amd


def load_dict(path):
    """(.zxd) -> (content, 128-byte shared table); layout docs/FORMAT.md §12.4"""
    raw = open(path, "rb").read()
    n = raw[6] | (raw[7] << 8)
    return raw[16:16 + n], raw[16 + n:16 + n + 128]


def read(rel):
    return open(os.path.join(GOLDEN, rel), "rb").read()


@pytest.fixture(scope="session")
def mockdev(ref):
    """The product's HOST sources over a mock device (host memory + the reference's Block API): tests/mock_device. -> CDLL"""
    import ctypes
    import subprocess
    d = os.path.join(ROOT, "tests", "mock_device")
    subprocess.run(["make", "-C", d], check=True, capture_output=True)
    L = ctypes.CDLL(os.path.join(d, "_bin", "libzxc_mockdev.so"))
    assert L.zxc_mi355x_device_count() == 1
    return L


def push_random_schedule(L, data, arc_for_decode, rng, bs, checksum):
    """drive both state machines with a random (in bytes, out bytes) per call — zero-size ins and outs included"""
    import ctypes as C
    import zxc_amd.api as api
    api._bind_pstream(L)
    o = api._CompressOpts(level=3, block_size=bs, checksum_enabled=int(checksum))
    cs = L.zxc_cstream_create(C.byref(o))
    src = C.create_string_buffer(data, max(len(data), 1))
    obuf = C.create_string_buffer(4 * bs + 64)
    blob = bytearray()
    off = stall = 0
    while off < len(data):
        n = min(rng.choice((0, 1, 7, bs - 1, bs, bs + 1, 3 * bs + 5, rng.randrange(4 * bs))), len(data) - off)
        inb = api._InBuf(C.addressof(src) + off, n, 0)
        while True:
            out = api._OutBuf(C.addressof(obuf), rng.choice((0, 1, 16, 37, bs, 4 * bs)), 0)
            r = L.zxc_cstream_compress(cs, C.byref(out), C.byref(inb))
            assert r >= 0, r
            blob += C.string_at(out.dst, out.pos)
            stall = stall + 1 if (out.pos == 0 and r > 0) else 0
            assert stall < 64
            if r == 0 and inb.pos == inb.size:
                break
        off += n
    while True:
        out = api._OutBuf(C.addressof(obuf), rng.choice((0, 1, 5, 12, 4 * bs)), 0)
        r = L.zxc_cstream_end(cs, C.byref(out))
        assert r >= 0, r
        blob += C.string_at(out.dst, out.pos)
        if r == 0:
            break
    L.zxc_cstream_free(cs)
    ds = L.zxc_dstream_create(C.byref(api._DecompressOpts(checksum_enabled=int(checksum))))
    comp = C.create_string_buffer(arc_for_decode, len(arc_for_decode))
    dec = bytearray()
    off = idle = 0
    while not L.zxc_dstream_finished(ds):
        n = min(rng.choice((0, 1, 3, 8, 100, bs // 2, 2 * bs, rng.randrange(1, 5 * bs))), len(arc_for_decode) - off)
        inb = api._InBuf(C.addressof(comp) + off, n, 0)
        out = api._OutBuf(C.addressof(obuf), rng.choice((0, 1, 53, bs - 1, bs, 4 * bs)), 0)
        r = L.zxc_dstream_decompress(ds, C.byref(out), C.byref(inb))
        assert r >= 0 and r == out.pos, (r, out.pos)
        dec += C.string_at(out.dst, out.pos)
        off += inb.pos
        idle = idle + 1 if (inb.pos == 0 and out.pos == 0) else 0
        assert idle < 200, "the decoder neither consumes nor produces"
    L.zxc_dstream_free(ds)
    return bytes(blob), bytes(dec), off
