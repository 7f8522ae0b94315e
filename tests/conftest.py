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
