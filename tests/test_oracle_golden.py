"""CPU tests (no GPU): pin the oracle (oracle/zxc_oracle.c) against every fixture the
reference's own tests hold for the decode path, and against the reference itself."""
import hashlib
import os
import random

import pytest

from conftest import GOLDEN, load_dict, read


def _valid_names():
    d = os.path.join(GOLDEN, "conformance", "valid")
    return sorted(f[:-4] for f in os.listdir(d) if f.endswith(".zxc"))


@pytest.mark.parametrize("name", _valid_names())
def test_conformance_valid(oracle, name):
    comp = read(f"conformance/valid/{name}.zxc")
    exp = read(f"conformance/valid/{name}.expected")
    d = dh = None
    if name.startswith("dict_"):
        zxd = "dict_http.zxd" if name.startswith("dict_http") else "dict_text.zxd"
        d, dh = load_dict(os.path.join(GOLDEN, "conformance", "valid", zxd))
    rc, out = oracle.decompress(comp, len(exp), checksum=True, dict_=d, dict_huf=dh)
    assert rc == len(exp)
    assert out == exp


def test_conformance_invalid_pinned_codes(oracle, manifest):
    for f, meta in manifest["conformance_invalid"].items():
        rc, _ = oracle.decompress(read(f"conformance/invalid/{f}"), 1 << 20, checksum=True)
        assert rc == meta["expect"], f


def test_format_golden(oracle, manifest):
    dicts = {"09_block_dict.zxc", "12_glo_huffman_dict.zxc"}
    for f, meta in manifest["format"].items():
        if f in dicts or meta["ref_rc"] < 0:
            continue  # dictionary archives need the generator's dictionary (reference tests/format)
        rc, out = oracle.decompress(read(f"format/{f}"), meta["decoded_size"])
        assert rc == meta["decoded_size"], f
        assert hashlib.sha256(out).hexdigest() == meta["decoded_sha256"], f


def test_synth_archives(oracle, manifest, synth_inputs):
    for name, meta in manifest["synth"].items():
        comp = read(f"synth/{name}.zxc")
        assert hashlib.sha256(comp).hexdigest() == meta["comp_sha256"]
        data = synth_inputs[meta["input"]]
        assert hashlib.sha256(data).hexdigest() == meta["data_sha256"], "generator drifted: " + name
        rc, out = oracle.decompress(comp, len(data), checksum=bool(meta["checksum"]))
        assert rc == len(data) and out == data, name


def test_seek_table_and_ranges(oracle, manifest, synth_inputs):
    rng = random.Random(1)
    for name, meta in manifest["synth"].items():
        if not meta["seekable"]:
            assert oracle.seek_table(read(f"synth/{name}.zxc")) is None
            continue
        comp = read(f"synth/{name}.zxc")
        data = synth_inputs[meta["input"]]
        t = oracle.seek_table(comp)
        assert t["total"] == len(data) and t["block_size"] == meta["block_size"]
        assert t["n_blocks"] == (len(data) + meta["block_size"] - 1) // meta["block_size"]
        for _ in range(6):
            a = rng.randrange(0, len(data))
            n = rng.randrange(1, len(data) - a + 1)
            rc, out = oracle.seekable_range(comp, a, n)
            assert rc == n and out == data[a:a + n]
        rc, _ = oracle.seekable_range(comp, len(data) - 1, 2)
        assert rc == -3  # ZXC_ERROR_SRC_TOO_SMALL past the end


def test_checksum_and_header_hashes_match_reference_bytes(oracle):
    # FORMAT.md §14 worked example: header CRC16 0x5B6E, RAW block CRC8 0x69, checksum 0x75A1BB90
    arc = bytes.fromhex("F52EB09C08138000000000000000 6E5B 0000000A00000069 48656C6C6F205A58430A 90BBA175"
                        "FF00000000000002 0A00000000000000 90BBA175".replace(" ", ""))
    rc, out = oracle.decompress(arc, 10, checksum=True)
    assert rc == 10 and out == b"Hello ZXC\n"
    assert oracle.lib.zxo_checksum32(b"Hello ZXC\n", 10) == 0x75A1BB90


def test_differential_vs_reference_on_mutations(oracle, ref, manifest):
    """Flip bytes in real archives: whenever both decoders accept, bytes must agree; error
    codes are compared and the agreement rate reported (phase-dependent codes may differ)."""
    rng = random.Random(7)
    agree = total = 0
    for name in ("mixed_384k_l3_b64k", "mixed_384k_l1_b64k", "text_200k_l3_b4k", "mixed_384k_l7_b64k"):
        comp = bytearray(read(f"synth/{name}.zxc"))
        size = manifest["synth"][name]["size"]
        for _ in range(60):
            m = bytearray(comp)
            for _ in range(rng.choice((1, 1, 2, 4))):
                m[rng.randrange(16, len(m) - 12)] ^= 1 << rng.randrange(8)
            a, ao = oracle.decompress(bytes(m), size)
            b, bo = ref.decompress(bytes(m), size)
            total += 1
            if a >= 0 and b >= 0:
                assert a == b and ao == bo
            agree += (a == b) or (a >= 0 and b >= 0)
    assert agree / total > 0.90, f"oracle/reference status agreement {agree}/{total}"
