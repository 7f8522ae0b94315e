"""The reference's OWN C test programs against libzxc_mi355x.so (VERDICT r2 missing #5 / next #6).

tests/c_abi/Makefile compiles /root/reference/conformance/test_conformance.c and the public-API cases of
/root/reference/tests/*.c (Block API, Buffer API, contexts, seekable, seekable MT) in place — with the reference's own
headers — and links them against the product library: programs written for the reference's zxc run unchanged against this
one. The binaries are built where /root/reference exists (__graft_entry__.build()) and travel to the GPU box."""
import os
import re
import subprocess
import sys

import pytest

from conftest import GOLDEN

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "c_abi", "_bin")
CONF = os.path.join(BIN, "zxc_conformance_test")
UNIT = os.path.join(BIN, "zxc_unit_subset")

# Cases of the reference's suite that cannot hold for a device-side codec, each with its reason (asserted to FAIL or PASS
# exactly as listed, so a change in either direction shows up):
KNOWN = {
}


def _built():
    if not (os.path.exists(CONF) and os.path.exists(UNIT)):
        if os.path.isdir("/root/reference/tests"):
            subprocess.run(["make", "-C", os.path.join(HERE, "c_abi"), "all"], check=True, stdout=subprocess.DEVNULL)
        else:
            pytest.skip("tests/c_abi/_bin not built (needs /root/reference at build time)")


def test_c_programs_link_against_the_product_library():
    """No GPU needed: the binaries exist, resolve libzxc_mi355x.so (nothing else of zxc), and list their cases."""
    _built()
    for exe in (CONF, UNIT):
        out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
        assert "libzxc_mi355x.so" in out and "libzxc_ref" not in out and "not found" not in out, out
    names = subprocess.run([UNIT, "--list"], capture_output=True, text=True, check=True).stdout.split()
    assert len(names) >= 94 and "test_block_api" in names and "test_seekable_mt_roundtrip" in names and "test_pstream_tiny_chunks" in names


def test_reference_unit_cases_pass_on_the_mock_device(ref):
    """The same 94 cases against the product's HOST sources over tests/mock_device (host memory; the two kernels' contracts met by
    the reference's Block API): the host logic — framing, batching, contexts, push streams, seekable ranges, error codes —
    without a GPU. The device itself: the two tests below."""
    _built()
    mock = os.path.join(os.path.dirname(UNIT), "zxc_unit_subset_mock")
    if not os.path.exists(mock):
        pytest.skip("tests/c_abi/_bin/zxc_unit_subset_mock not built (needs /root/reference at build time)")
    r = subprocess.run([mock], capture_output=True, text=True, timeout=900)
    res = dict(re.findall(r"^RESULT (\S+) (PASS|FAIL)$", r.stdout, flags=re.M))
    assert len(res) >= 94 and all(v == "PASS" for v in res.values()), ({k: v for k, v in res.items() if v != "PASS"}, r.stdout[-3000:])


def test_reference_python_wrapper_passes_its_own_tests_on_the_mock_device(ref):
    """The reference's Python wrapper (its C extension compiled in place from wrappers/python/src/zxc/_zxc.c, its package imported
    from where it lies) linked against this library's host sources over the mock device: the wrapper's OWN test suite
    (wrappers/python/tests: buffer, dict, io adapters, push streams, seekable, FILE* streams) passes — a language binding keeps
    working after the relink."""
    _built()
    ext_dir = os.path.join(os.path.dirname(UNIT), "pywrap")
    if not (os.path.isdir(ext_dir) and os.path.isdir("/root/reference/wrappers/python/tests")):
        pytest.skip("tests/c_abi/_bin/pywrap not built / the reference's wrapper tests are not here")
    r = subprocess.run([sys.executable, os.path.join(HERE, "c_abi", "run_ref_pytests.py"), ext_dir], capture_output=True, text=True, timeout=900)
    m = re.search(r"(\d+) passed", r.stdout)
    assert r.returncode == 0 and m and int(m.group(1)) >= 80 and "failed" not in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def _cli_scenario(cli, ref, tmp_path, exact):
    """compress (seekable, checksums: the tool's defaults), test, list, decompress, pipes, train a dictionary, use it, miss it"""
    import random
    rng = random.Random(3)
    words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randrange(3, 10))) for _ in range(200)]
    text = b" ".join(rng.choice(words) for _ in range(300000))
    (tmp_path / "a.txt").write_bytes(text)
    for i in range(40):
        (tmp_path / f"s{i}.json").write_bytes(b"".join(b'{"id": %d, "user": "user%d", "status": "active", "tags": ["alpha", "beta"]}\n' % (j, j % 97)
                                                        for j in range(i * 30, (i + 1) * 30)))

    def run(*args, stdin=None):
        return subprocess.run([cli, *args], cwd=tmp_path, capture_output=True, timeout=300, input=stdin)
    r = run("-3", "-S", "a.txt", "-o", "a.zxc")
    assert r.returncode == 0, r.stderr[-500:]
    arc = (tmp_path / "a.zxc").read_bytes()
    assert ref.decompress(arc, len(text), checksum=True) == (len(text), text)
    if exact:  # (on the mock device the block codec is the reference's: the file is the reference's zxc_compress of the same bytes)
        assert arc == ref.compress(text, 3, 512 * 1024, True, True)
    r = run("-t", "a.zxc")
    assert r.returncode == 0 and b"OK" in r.stdout + r.stderr
    r = run("-l", "a.zxc")
    assert r.returncode == 0 and b"a.zxc" in r.stdout + r.stderr
    assert run("-d", "a.zxc", "-o", "a.out").returncode == 0 and (tmp_path / "a.out").read_bytes() == text
    z = run("-z", "-N", stdin=text)
    assert z.returncode == 0 and ref.decompress(z.stdout, len(text)) == (len(text), text)
    d = run("-d", stdin=z.stdout)
    assert d.returncode == 0 and d.stdout == text
    bad = bytearray(arc)
    bad[len(bad) // 2] ^= 0x20
    (tmp_path / "bad.zxc").write_bytes(bytes(bad))
    assert run("-t", "bad.zxc").returncode != 0
    r = run("--train", *[f"s{i}.json" for i in range(40)], "-o", "d.zxd")
    assert r.returncode == 0 and (tmp_path / "d.zxd").stat().st_size > 16 + 128, r.stderr[-500:]
    assert run("-5", "-B", "4K", "-D", "d.zxd", "s1.json", "-o", "s1.zxc").returncode == 0
    assert run("-5", "-B", "4K", "s1.json", "-o", "s1_plain.zxc").returncode == 0
    assert (tmp_path / "s1.zxc").stat().st_size < (tmp_path / "s1_plain.zxc").stat().st_size  # (the dictionary pays)
    assert run("-d", "-D", "d.zxd", "s1.zxc", "-o", "s1.out").returncode == 0
    assert (tmp_path / "s1.out").read_bytes() == (tmp_path / "s1.json").read_bytes()
    r = run("-d", "s1.zxc", "-o", "s1.none")
    assert r.returncode != 0 and b"DICT_REQUIRED" in r.stdout + r.stderr


def _cli(name):
    _built()
    exe = os.path.join(os.path.dirname(UNIT), name)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (needs /root/reference at build time)")
    return exe


def test_reference_cli_on_the_mock_device(ref, tmp_path):
    """The reference's command-line tool (src/cli/main.c, compiled in place, public API only) linked against this library's host
    sources over the mock device."""
    _cli_scenario(_cli("zxc_cli_mock"), ref, tmp_path, exact=True)


@pytest.mark.gpu
def test_reference_cli_on_the_device(ref, tmp_path):
    """The same tool linked against libzxc_mi355x.so: the reference's CLI compresses, tests, lists, decompresses, trains and uses a
    dictionary on the GPU; the unmodified reference library reads what it wrote."""
    _cli_scenario(_cli("zxc_cli"), ref, tmp_path, exact=False)


FUZZ = ("roundtrip", "decompress", "seekable", "pstream", "dict")


def _run_fuzz(prefix, iters, seed):
    _built()
    for name in FUZZ:
        exe = os.path.join(os.path.dirname(UNIT), prefix + name)
        if not os.path.exists(exe):
            pytest.skip(f"{exe} not built (needs /root/reference at build time)")
        r = subprocess.run([exe, str(iters), str(seed)], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and f"FUZZ OK {iters} inputs" in r.stdout, (name, r.returncode, r.stdout[-500:], r.stderr[-1500:])


def test_reference_fuzz_harnesses_on_the_mock_device(ref):
    """The reference's own libFuzzer harnesses (tests/fuzz_*.c: round trip, decoder robustness, seekable, push streams, dictionaries),
    asserts on, behind a deterministic input generator (tests/c_abi/fuzz_driver.c: random / skewed / repetitive data and stomped
    valid archives), against the host sources over the mock device."""
    _run_fuzz("fuzz_mock_", 600, 11)


@pytest.mark.gpu
def test_reference_fuzz_harnesses_pass():
    """The same harnesses against the product library on the GPU."""
    _run_fuzz("fuzz_", 500, 5)


@pytest.mark.gpu
def test_reference_conformance_program_passes():
    """reference conformance/test_conformance.c: every valid vector decodes to its .expected bytes, every invalid vector is
    rejected with the code pinned at :228-249 — through this library, on the GPU."""
    _built()
    r = subprocess.run([CONF, os.path.join(GOLDEN, "conformance")], capture_output=True, text=True, timeout=600)
    tail = r.stdout[-3000:]
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "FAIL" not in r.stdout, tail
    assert len(re.findall(r"PASS", r.stdout)) >= 50, tail


@pytest.mark.gpu
def test_reference_unit_cases_pass():
    """The public-API cases of reference tests/test_main.c (Buffer / Block / context / seekable / seekable-MT / push streaming / static contexts / dictionaries incl. training / FILE* streams), compiled from
    the reference's own sources, all pass against this library."""
    _built()
    r = subprocess.run([UNIT], capture_output=True, text=True, timeout=1800)
    res = dict(re.findall(r"^RESULT (\S+) (PASS|FAIL)$", r.stdout, flags=re.M))
    assert len(res) >= 94, r.stdout[-3000:] + r.stderr[-2000:]
    bad = {k: v for k, v in res.items() if (v == "FAIL") != (k in KNOWN)}
    assert not bad, (bad, r.stdout[-6000:])
