"""The HOST sources of the library under AddressSanitizer + UBSan + LeakSanitizer and under ThreadSanitizer, on the mock device
(tests/mock_device: host memory, the reference's Block API in place of the two kernels), with 1 MiB pieces so that every call of any
size runs the piece pipeline with its producer threads: a four-thread stress program (frames, push streams in odd chunkings,
seekable ranges incl. _mt; tests/mock_device/stress.c) and the reference's own 94 unit cases. What the device side cannot show on
a box without a GPU — races between the pipeline's threads, leaks of arenas / events / contexts, out-of-bounds staging — shows here."""
import os
import subprocess

import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "tests", "mock_device", "_bin")


@pytest.fixture(scope="module")
def san(ref):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "mock_device"), "san"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer builds failed (no libasan / libtsan here?): " + r.stderr[-300:])
    return dict(os.environ, ZXC_MI355X_FRAME_BATCH_MIB="1", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
                TSAN_OPTIONS="halt_on_error=1", ZXC_MOCK_REF_SO=os.path.join(ROOT, "oracle", "_ref", "libzxc_ref.so"))


def _clean(r, what):
    bad = [k for k in ("ERROR: AddressSanitizer", "runtime error:", "LeakSanitizer", "WARNING: ThreadSanitizer") if k in r.stderr]
    assert r.returncode == 0 and not bad, (what, r.returncode, bad, r.stdout[-300:], r.stderr[-3000:])


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_four_thread_stress(san, kind):
    r = subprocess.run([os.path.join(BIN, "stress_" + kind), "4"], capture_output=True, text=True, timeout=900, env=san)
    _clean(r, "stress_" + kind)
    assert "STRESS OK" in r.stdout


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_reference_unit_cases(san, kind):
    exe = os.path.join(BIN, "unit_" + kind)
    if not os.path.exists(exe):
        pytest.skip("needs /root/reference at build time")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=1500, env=san)
    _clean(r, "unit_" + kind)
    assert "SUMMARY ran 94 failed 0" in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("name", ["roundtrip", "decompress", "seekable", "pstream", "dict"])
def test_reference_fuzz_harnesses_under_asan(san, name):
    """the reference's libFuzzer harnesses behind tests/c_abi/fuzz_driver.c (random / skewed / repetitive data, stomped and truncated
    valid archives): the host's parsers — frame walk, seek table, push-stream state machines, .zxd — must stay in bounds"""
    exe = os.path.join(BIN, "fuzz_asan_" + name)
    if not os.path.exists(exe):
        pytest.skip("needs /root/reference at build time")
    r = subprocess.run([exe, "1500", "3"], capture_output=True, text=True, timeout=900, env=dict(san, ZXC_MI355X_FRAME_BATCH_MIB=""))
    _clean(r, "fuzz_asan_" + name)
    assert "FUZZ OK 1500 inputs" in r.stdout
