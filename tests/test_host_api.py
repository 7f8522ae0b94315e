"""CPU tests of the product library's host side (no GPU compute): it loads, exports every
symbol include/*.h declares, parses containers like the reference, and fails loudly —
never falls back to a CPU decoder — when there is no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, read


def test_exports_every_declared_symbol(product):
    L = product.lib()
    declared = set()
    for h in os.listdir(os.path.join(ROOT, "include")):
        txt = open(os.path.join(ROOT, "include", h)).read()
        declared |= set(re.findall(r"ZXC_EXPORT[^;(]*?\b(zxc_\w+)\s*\(", txt))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(L, name), f"include/ declares {name} but libzxc_mi355x.so does not export it"


def test_exports_the_whole_public_api_of_the_reference_and_what_its_wrappers_call(product):
    """every function the reference's public headers export (67) — and with them every C function its five language wrappers
    (go, rust, python, nodejs, wasm: they bind the C API, push streaming included) call — is exported here under the same name"""
    ref_inc = "/root/reference/include"
    if not os.path.isdir(ref_inc):
        pytest.skip("/root/reference is not here")
    L = product.lib()
    public = set()
    for h in os.listdir(ref_inc):
        public |= set(re.findall(r"ZXC_EXPORT[^;(]*?\b(zxc_\w+)\s*\(", open(os.path.join(ref_inc, h)).read()))
    assert len(public) == 67
    missing = sorted(n for n in public if not hasattr(L, n))
    assert not missing, missing
    used = set()
    for rel in ("python/src/zxc/_zxc.c", "nodejs/src/zxc_addon.cc", "rust/zxc-sys/src/lib.rs", "go/zxc.go", "go/zxc_stream.go", "wasm/zxc_wasm.js"):
        f = os.path.join("/root/reference/wrappers", rel)
        if os.path.exists(f):
            used |= set(re.findall(r"\b_?(zxc_[a-z0-9_]+)\b", open(f, errors="replace").read())) & public
    assert len(used) >= 30 and all(hasattr(L, n) for n in used)


def test_no_oracle_or_reference_linked(product):
    """The product must not link the checkers."""
    import subprocess
    out = subprocess.run(["ldd", product.lib_path()], capture_output=True, text=True).stdout
    assert "oracle" not in out and "zxc_ref" not in out
    syms = subprocess.run(["nm", "-D", product.lib_path()], capture_output=True, text=True).stdout
    assert "zxo_" not in syms


def test_opts_layout_and_misc(product):
    L = product.lib()
    L.zxc_compress_opts_size.restype = C.c_size_t
    L.zxc_decompress_opts_size.restype = C.c_size_t
    assert L.zxc_compress_opts_size() == 64 and L.zxc_decompress_opts_size() == 48  # include/zxc_opts.h:105-111
    assert product.error_name(-9) == "ZXC_ERROR_BAD_OFFSET"
    assert product.error_name(-100) == "ZXC_ERROR_GPU_UNAVAILABLE"
    L.zxc_version_string.restype = C.c_char_p
    assert L.zxc_version_string() == b"0.13.3"
    assert (L.zxc_min_level(), L.zxc_default_level(), L.zxc_max_level()) == (1, 3, 7)


def test_compress_bound_matches_reference(product, ref):
    for n in (0, 1, 4095, 4096, 65536, 1 << 20, 211947520):
        assert product.lib().zxc_compress_bound(n) == ref.lib.zxc_compress_bound(n)


def test_block_bounds_and_dict_id_match_reference(product, ref):
    """zxc_compress_block_bound / zxc_decompress_block_bound (src/lib/zxc_common.c:873-902) and zxc_get_dict_id
    (src/lib/zxc_dispatch.c:1234-1242) are pure host arithmetic: same numbers as the reference."""
    import oracle_py
    P = oracle_py.bind_block_api(C.CDLL(product.lib_path()))
    Rl = oracle_py.bind_block_api(ref.lib)
    for n in (0, 1, 4095, 4096, 65536, (1 << 21) - 1, 1 << 21, (1 << 21) + 1, 1 << 30):
        assert P.zxc_compress_block_bound(n) == Rl.zxc_compress_block_bound(n), n
        assert P.zxc_decompress_block_bound(n) == Rl.zxc_decompress_block_bound(n), n
    for f in sorted(os.listdir(os.path.join(GOLDEN, "conformance", "valid"))):
        if f.endswith(".zxc"):
            b = read(f"conformance/valid/{f}")
            assert P.zxc_get_dict_id(b, len(b)) == Rl.zxc_get_dict_id(b, len(b)), f
            assert (P.zxc_get_dict_id(b, len(b)) != 0) == f.startswith("dict_")
    assert P.zxc_get_dict_id(b"\0" * 16, 16) == 0 and P.zxc_get_dict_id(b"", 0) == 0


def test_inplace_bound_matches_reference(product, ref, manifest):
    """zxc_decompress_inplace_bound (src/lib/zxc_dispatch.c:1129-1145) is host arithmetic over header + footer: the same
    number as the reference for every golden archive, 0 for garbage; argument errors of zxc_decompress_inplace alike."""
    P, R = C.CDLL(product.lib_path()), ref.lib
    for L in (P, R):
        L.zxc_decompress_inplace_bound.restype = C.c_size_t
        L.zxc_decompress_inplace_bound.argtypes = [C.c_char_p, C.c_size_t]
        L.zxc_decompress_inplace.restype = C.c_int64
        L.zxc_decompress_inplace.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    for name in manifest["synth"]:
        b = read(f"synth/{name}.zxc")
        assert P.zxc_decompress_inplace_bound(b, len(b)) == R.zxc_decompress_inplace_bound(b, len(b)) > 0, name
    for f in manifest["conformance_invalid"]:
        b = read(f"conformance/invalid/{f}")
        assert P.zxc_decompress_inplace_bound(b, len(b)) == R.zxc_decompress_inplace_bound(b, len(b)), f
    b = read("synth/mixed_384k_l3_b64k.zxc")
    buf = C.create_string_buffer(len(b) + 100)
    C.memmove(C.addressof(buf) + 100, b, len(b))
    assert P.zxc_decompress_inplace(buf, len(b) + 100, len(b), None) == R.zxc_decompress_inplace(buf, len(b) + 100, len(b), None) == -2
    assert P.zxc_decompress_inplace(buf, 10, len(b), None) == R.zxc_decompress_inplace(buf, 10, len(b), None) == -12


def test_zxd_container_helpers_match_reference(product, ref):
    """zxc_dict_id / zxc_dict_load / zxc_dict_save / zxc_dict_get_id / zxc_dict_huf (reference src/lib/zxc_dict.c:35-205):
    same ids, same bytes, same error codes as the unmodified reference on the conformance .zxd files and on damaged ones."""
    def bind(L):
        L.zxc_dict_id.restype = C.c_uint32
        L.zxc_dict_id.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.zxc_dict_get_id.restype = C.c_uint32
        L.zxc_dict_get_id.argtypes = [C.c_char_p, C.c_size_t]
        L.zxc_dict_save_bound.restype = C.c_size_t
        L.zxc_dict_save_bound.argtypes = [C.c_size_t]
        L.zxc_dict_save.restype = C.c_int64
        L.zxc_dict_save.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_void_p, C.c_size_t]
        L.zxc_dict_load.restype = C.c_int
        L.zxc_dict_load.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        L.zxc_dict_huf.restype = C.c_void_p
        L.zxc_dict_huf.argtypes = [C.c_char_p, C.c_size_t]
        return L
    P, R = bind(C.CDLL(product.lib_path())), bind(ref.lib)
    for name in ("dict_http.zxd", "dict_text.zxd"):
        zxd = read(f"conformance/valid/{name}")
        n = zxd[6] | (zxd[7] << 8)
        content, huf = zxd[16:16 + n], zxd[16 + n:16 + n + 128]
        assert P.zxc_dict_id(content, n, huf) == R.zxc_dict_id(content, n, huf) == P.zxc_dict_get_id(zxd, len(zxd))
        assert P.zxc_dict_id(content, n, None) == R.zxc_dict_id(content, n, None)
        assert P.zxc_dict_save_bound(n) == R.zxc_dict_save_bound(n) == len(zxd)
        out = C.create_string_buffer(len(zxd))
        assert P.zxc_dict_save(content, n, huf, out, len(zxd)) == len(zxd) and out.raw == zxd
        assert P.zxc_dict_save(content, n, huf, out, len(zxd) - 1) == R.zxc_dict_save(content, n, huf, out, len(zxd) - 1) == -2
        for mutate in (None, 0, 4, 6, 9, 14, 20, len(zxd) - 1):
            m = bytearray(zxd)
            if mutate is not None:
                m[mutate] ^= 0x10
            res = []
            for L in (P, R):
                cp, cs, hp, did = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_uint32()
                rc = L.zxc_dict_load(bytes(m), len(m), C.byref(cp), C.byref(cs), C.byref(hp), C.byref(did))
                res.append((rc, cs.value if rc == 0 else 0, did.value if rc == 0 else 0))
            assert res[0] == res[1], (name, mutate, res)
        assert P.zxc_dict_load(zxd[:10], 10, C.byref(C.c_void_p()), C.byref(C.c_size_t()), None, None) == -3
        assert bool(P.zxc_dict_huf(zxd, len(zxd))) and not P.zxc_dict_huf(b"\0" * 200, 200)


def test_container_errors_need_no_gpu(product, manifest):
    """Header-level rejections happen on the host before any device work, with the
    reference's pinned codes (conformance/test_conformance.c:228-249)."""
    host_level = {"all_0xff_garbage", "bad_block_size_field", "bad_checksum_algo", "bad_header_crc", "bad_magic",
                  "bad_version", "dict_required", "magic_then_zeros", "too_short_4bytes",
                  "truncated_header_only", "zero_length"}
    for f, meta in manifest["conformance_invalid"].items():
        if f[:-4] in host_level:
            rc, _ = product.decompress(read(f"conformance/invalid/{f}"), 1 << 20, raise_on_error=False)
            assert rc == meta["expect"], f


def test_seekable_handle_matches_oracle(product, oracle, manifest):
    for name, meta in manifest["synth"].items():
        comp = read(f"synth/{name}.zxc")
        if not meta["seekable"]:
            with pytest.raises(product.ZxcError):
                product.Seekable(comp)
            continue
        s = product.Seekable(comp)
        t = oracle.seek_table(comp)
        assert s.num_blocks == t["n_blocks"] and s.decompressed_size == t["total"]
        assert [s.block_comp_size(i) for i in range(s.num_blocks)] == t["comp_sizes"]
        jobs = s.plan()
        assert list(jobs["comp_off"]) == t["comp_offsets"][:-1]
        assert list(jobs["comp_size"]) == t["comp_sizes"]
        assert all(int(o) % 16 == 0 for o in jobs["out_off"])
        assert int(jobs["out_len"].sum()) == t["total"]
        assert product.get_decompressed_size(comp) == t["total"]
        s.close()


def test_seek_table_writer_matches_reference_bytes(product):
    # the SEK block of a stored archive must be reproduced byte for byte
    comp = read("synth/seek_70001_l3_b16k.zxc")
    s = product.Seekable(comp)
    n = s.num_blocks
    sizes = (C.c_uint32 * n)(*[s.block_comp_size(i) for i in range(n)])
    L = product.lib()
    L.zxc_seek_table_size.restype = C.c_size_t
    L.zxc_write_seek_table.restype = C.c_int64
    tot = L.zxc_seek_table_size(n)
    buf = C.create_string_buffer(tot)
    assert L.zxc_write_seek_table(buf, C.c_size_t(tot), sizes, C.c_uint32(n)) == tot
    assert buf.raw == comp[len(comp) - 12 - tot:len(comp) - 12]


@pytest.mark.skipif(os.environ.get("HIP_VISIBLE_DEVICES", "") != "" or os.path.exists("/dev/kfd"),
                    reason="a GPU is visible")
def test_fails_loudly_without_gpu(product):
    comp = read("synth/lorem_100k_l3_b64k.zxc")
    rc, _ = product.decompress(comp, raise_on_error=False)
    assert rc == -100  # ZXC_ERROR_GPU_UNAVAILABLE, not a CPU fallback
    s = product.Seekable(comp)
    rc, _ = s.decompress_range(0, 1000, raise_on_error=False)
    assert rc == -100
