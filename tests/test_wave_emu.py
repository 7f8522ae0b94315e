"""The HIP decode kernels, compiled unchanged for the CPU wave emulator (tests/wave_emu: 64 fibers in
lock-step, cross-lane operations as rendezvous, the ring's ds_or / store / load points modelled with the
wave's LDS ordering rules), must be bit-exact against the oracle. Runs without a GPU: this is where the
kernel LOGIC is checked on every commit; the -m gpu tests check the same on the hardware."""
import os
import random
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_dict, read

sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emu"))


@pytest.fixture(scope="module")
def emu():
    import emu_py
    return emu_py.Emu()


def _check_archive(emu, oracle, comp, dict_=None, dict_huf=None, verify=False):
    import emu_py
    jobs, bs, ck, total = emu_py.frame_jobs(comp)
    rc, want = oracle.decompress(comp, total, checksum=verify, dict_=dict_, dict_huf=dict_huf)
    assert rc == total
    st, out = emu.decode_jobs(comp, jobs, total, bs, verify_trailer=ck and verify, dict_=dict_, dict_huf=dict_huf)
    assert (st == jobs["out_len"].astype(np.int32)).all(), st[:8]
    assert out == want


def _valid_names():
    d = os.path.join(GOLDEN, "conformance", "valid")
    return sorted(f[:-4] for f in os.listdir(d) if f.endswith(".zxc"))


@pytest.mark.parametrize("name", _valid_names())
def test_conformance_valid_on_emulator(emu, oracle, name):
    comp = read(f"conformance/valid/{name}.zxc")
    d = dh = None
    if name.startswith("dict_"):
        zxd = "dict_http.zxd" if name.startswith("dict_http") else "dict_text.zxd"
        d, dh = load_dict(os.path.join(GOLDEN, "conformance", "valid", zxd))
    _check_archive(emu, oracle, comp, d, dh, verify=True)


def test_synth_archives_on_emulator(emu, oracle, manifest):
    n = 0
    for name, meta in manifest["synth"].items():
        if meta["size"] > 400_000:
            continue  # (the 2 MiB single-block archive takes 7 s here; the GPU suite has it)
        _check_archive(emu, oracle, read(f"synth/{name}.zxc"), verify=bool(meta["checksum"]))
        n += 1
    assert n >= 12


def test_mutated_blocks_on_emulator_match_oracle(emu, oracle):
    """Per-block differential fuzz: same status code, same bytes (the executor's error paths)."""
    import emu_py
    rng = random.Random(5)
    for name, rounds in (("text_200k_l3_b4k", 6), ("seek_70001_l3_b16k", 10), ("mixed_384k_l1_b64k", 2),
                         ("mixed_384k_l3_b64k", 2), ("mixed_384k_l7_b64k", 2)):
        comp = read(f"synth/{name}.zxc")
        jobs, bs, ck, total = emu_py.frame_jobs(comp)
        for _ in range(rounds):
            m = bytearray(comp)
            for j in jobs:  # one to three bit flips inside every block's payload
                for _ in range(rng.choice((1, 1, 2, 3))):
                    m[int(j["comp_off"]) + 8 + rng.randrange(int(j["comp_size"]) - 8)] ^= 1 << rng.randrange(8)
            m = bytes(m)
            st, out = emu.decode_jobs(m, jobs, total, bs)
            for i, j in enumerate(jobs):
                blk = m[int(j["comp_off"]):int(j["comp_off"]) + int(j["comp_size"])]
                rc, want = oracle.decode_block(blk, bs)
                assert st[i] == rc, (name, i, st[i], rc)
                if rc >= 0:
                    n = min(rc, int(j["out_len"]))
                    o = int(j["out_off"])
                    assert out[o:o + n] == want[:n], (name, i)


def test_levels_and_block_sizes_on_emulator(emu, oracle, ref):
    """Reference-encoded archives of every level at small block sizes (cheap here): the GLO / GHI parse,
    RLE and PivCo sections in front of the ds_or executor, ragged last blocks."""
    from zxc_amd import corpus
    data = corpus.synth_silesia(96 << 10, seed=4) + corpus.small_offset_pattern(3000) + bytes(5000) + corpus.period300(9000)
    for level, bs in ((1, 4096), (2, 16384), (3, 8192), (4, 4096), (5, 16384), (6, 8192), (7, 16384)):
        _check_archive(emu, oracle, ref.compress(data, level, bs, True, level == 4), verify=True)
