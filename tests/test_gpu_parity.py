"""GPU parity tests (run with -m gpu on an MI355X): the HIP decode path, called through the
C-ABI of libzxc_mi355x.so, must be bit-exact against the oracle / golden fixtures."""
import ctypes as C
import hashlib
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN, load_dict, read

pytestmark = pytest.mark.gpu

# sections the device path does not decode yet (reported as ZXC_ERROR_GPU_UNSUPPORTED = -101)
UNSUPPORTED = -101


def _valid_names():
    d = os.path.join(GOLDEN, "conformance", "valid")
    return sorted(f[:-4] for f in os.listdir(d) if f.endswith(".zxc"))


@pytest.fixture(scope="module")
def gpu(product):
    assert product.lib().zxc_mi355x_device_count() >= 1, "no HIP device"
    product.lib().zxc_mi355x_set_device(0)
    return product


@pytest.mark.parametrize("name", _valid_names())
def test_conformance_valid(gpu, oracle, name):
    comp = read(f"conformance/valid/{name}.zxc")
    exp = read(f"conformance/valid/{name}.expected")
    d = dh = None
    if name.startswith("dict_"):
        assert gpu.decompress(comp, len(exp), raise_on_error=False)[0] == -15  # ZXC_ERROR_DICT_REQUIRED without it
        zxd = "dict_http.zxd" if name.startswith("dict_http") else "dict_text.zxd"
        d, dh = load_dict(os.path.join(GOLDEN, "conformance", "valid", zxd))
        assert gpu.decompress(comp, len(exp), raise_on_error=False, dict_=d[:-1] + b"?", dict_huf=dh)[0] == -16  # mismatch
    rc, out = gpu.decompress(comp, len(exp), checksum=True, raise_on_error=False, dict_=d, dict_huf=dh)
    assert rc == len(exp) and out == exp
    if name == "dict_seekable_l7":
        s = gpu.Seekable(comp)
        assert s.decompress_range(0, 10, raise_on_error=False)[0] == -15
        assert s.set_dict(d, dh) == 0
        assert s.decompress_range(0, len(exp)) == exp
        assert s.decompress_range(len(exp) // 3, len(exp) // 2) == exp[len(exp) // 3: len(exp) // 3 + len(exp) // 2]


def test_conformance_invalid(gpu, manifest):
    for f, meta in manifest["conformance_invalid"].items():
        rc, _ = gpu.decompress(read(f"conformance/invalid/{f}"), 1 << 20, checksum=True, raise_on_error=False)
        assert rc == meta["expect"], f


def test_format_golden(gpu, manifest):
    for f, meta in manifest["format"].items():
        if meta["ref_rc"] < 0:
            continue
        rc, out = gpu.decompress(read(f"format/{f}"), meta["decoded_size"], raise_on_error=False)
        if f in ("09_block_dict.zxc", "12_glo_huffman_dict.zxc"):
            assert rc == -15  # needs the generator's dictionary (reference tests/format); rejected loudly
            continue
        assert rc == meta["decoded_size"], f
        assert hashlib.sha256(out).hexdigest() == meta["decoded_sha256"], f


def test_synth_archives_all_levels(gpu, manifest, synth_inputs):
    seen_ok = 0
    for name, meta in manifest["synth"].items():
        comp = read(f"synth/{name}.zxc")
        data = synth_inputs[meta["input"]]
        rc, out = gpu.decompress(comp, len(data), checksum=bool(meta["checksum"]), raise_on_error=False)
        assert rc == len(data), (name, rc)
        assert out == data, name
        seen_ok += 1
    assert seen_ok >= 16


def test_seekable_ranges(gpu, manifest, synth_inputs):
    rng = random.Random(3)
    for name, meta in manifest["synth"].items():
        if not meta["seekable"]:
            continue
        comp = read(f"synth/{name}.zxc")
        data = synth_inputs[meta["input"]]
        s = gpu.Seekable(comp)
        assert s.decompress_range(0, len(data)) == data
        assert s.decompress_range(0, len(data), n_threads=8) == data
        for _ in range(5):
            a = rng.randrange(0, len(data))
            n = rng.randrange(1, len(data) - a + 1)
            assert s.decompress_range(a, n) == data[a:a + n], (name, a, n)
        rc, _ = s.decompress_range(len(data) - 1, 2, raise_on_error=False)
        assert rc == -3
        s.close()


def test_mutated_blocks_match_oracle_exactly(gpu, oracle, manifest):
    """Per-block differential fuzz: same accept/reject decision, same error code, same bytes."""
    rng = random.Random(11)
    for name in ("mixed_384k_l3_b64k", "mixed_384k_l1_b64k", "text_200k_l3_b4k", "period300_150k_l5_b64k",
                 "mixed_384k_l7_b64k", "mixed_384k_l6_b64k"):
        comp = read(f"synth/{name}.zxc")
        size = manifest["synth"][name]["size"]
        for _ in range(40):
            m = bytearray(comp)
            for _ in range(rng.choice((1, 1, 2, 3))):
                m[rng.randrange(24, len(m) - 40)] ^= 1 << rng.randrange(8)
            a, ao = oracle.decompress(bytes(m), size)
            b, bo = gpu.decompress(bytes(m), size, raise_on_error=False)
            assert a == b, (name, a, b)
            if a >= 0:
                assert ao == bo


def test_large_corpus_roundtrip_properties(gpu, ref):
    """Full-size property test: 32 MiB silesia-like corpus through the reference encoder, decoded on
    the GPU, sha256 must match the generator's bytes (size-independent round-trip property)."""
    from zxc_amd import corpus
    data = corpus.synth_silesia(32 << 20, seed=0)
    # 4 KiB blocks: 8192 blocks in one launch, more than one round of resident workgroups, so the
    # heaviest-first launch order (zxc_order_* kernels) is on
    for level, bs in ((1, 65536), (3, 65536), (5, 65536), (6, 65536), (3, 4096)):
        comp = ref.compress(data, level, bs, True, False)
        s = gpu.Seekable(comp)
        out = s.decompress_range(0, len(data))
        assert hashlib.sha256(out).digest() == hashlib.sha256(data).digest(), level
        s.close()


def test_checksum_verification(gpu, oracle, manifest):
    """Device rapidhash: a flipped payload bit must be caught as BAD_CHECKSUM (-7) when verification is
    requested, exactly like the oracle; the footer's global hash is checked too."""
    comp = bytearray(read("synth/text_300k_l1_b128k_ck.zxc"))
    size = manifest["synth"]["text_300k_l1_b128k_ck"]["size"]
    rng = random.Random(2)
    for _ in range(12):
        m = bytearray(comp)
        m[rng.randrange(40, len(m) - 40)] ^= 1 << rng.randrange(8)
        a, _ = oracle.decompress(bytes(m), size, checksum=True)
        b, _ = gpu.decompress(bytes(m), size, checksum=True, raise_on_error=False)
        assert a == b
    m = bytearray(comp)
    m[-1] ^= 0x40  # global hash in the footer
    assert gpu.decompress(bytes(m), size, checksum=True, raise_on_error=False)[0] == -7
    assert gpu.decompress(bytes(m), size, checksum=False) is not None


def test_concurrent_launches_on_two_streams(gpu, ref):
    """Two streams decode level-6 archives (scratch slots, PivCo) and long launches (ordered dispatch) at the
    same time: the lock-free scratch pool and the per-stream order buffers must keep them apart."""
    import torch
    from zxc_amd import corpus
    dev = torch.device("cuda", 0)
    data = corpus.synth_silesia(24 << 20, seed=9)
    work = []
    for level, bs in ((6, 65536), (3, 4096)):  # 384 PivCo blocks; 6144 blocks (> one round of workgroups)
        comp = ref.compress(data, level, bs, True, False)
        s = gpu.Seekable(comp)
        jobs = s.plan()
        d_comp = torch.frombuffer(bytearray(comp) + bytearray(64), dtype=torch.uint8).to(dev)
        d_jobs = torch.frombuffer(bytearray(jobs.tobytes()), dtype=torch.uint8).to(dev)
        d_out = torch.zeros(len(data) + 256, dtype=torch.uint8, device=dev)
        d_st = torch.zeros(jobs.size, dtype=torch.int32, device=dev)
        work.append((bs, jobs, d_comp, d_jobs, d_out, d_st, torch.cuda.Stream(device=dev)))
        s.close()
    want = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    for _ in range(6):
        for bs, jobs, d_comp, d_jobs, d_out, d_st, st in work:
            d_out.zero_()
        torch.cuda.synchronize()
        for rep in range(3):
            for bs, jobs, d_comp, d_jobs, d_out, d_st, st in work:
                gpu.decode_blocks_device(d_comp.data_ptr(), d_jobs.data_ptr(), jobs.size, d_out.data_ptr(),
                                         d_st.data_ptr(), bs, False, st.cuda_stream)
        torch.cuda.synchronize()
        for bs, jobs, d_comp, d_jobs, d_out, d_st, st in work:
            assert (d_st.cpu().numpy() == jobs["out_len"].astype(np.int32)).all(), bs
            assert torch.equal(d_out[:len(data)], want), bs
