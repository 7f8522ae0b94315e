"""Dictionary training (include/zxc_dict.h:146-191). zxc_train_dict is host arithmetic on the samples, restated from the reference:
its output must be the UNMODIFIED reference's bytes (no device involved). zxc_train_dict_huf / zxc_dict_train run the samples'
slices through the block encoder with the dictionary — on the mock device here (the reference's Block API under the product's host
code), on the MI355X in the gpu-marked tests — and must give a table BOTH libraries accept and that pays on small blocks."""
import ctypes as C
import random

import pytest

import zxc_amd.api as api


def _bind(L):
    L.zxc_train_dict.restype = C.c_int64
    L.zxc_train_dict.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_void_p, C.c_size_t]
    L.zxc_train_dict_huf.restype = C.c_int
    L.zxc_train_dict_huf.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.zxc_dict_train.restype = C.c_int64
    L.zxc_dict_train.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_void_p, C.c_size_t]
    L.zxc_dict_load.restype = C.c_int
    L.zxc_dict_load.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
    L.zxc_dict_id.restype = C.c_uint32
    L.zxc_dict_id.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    return L


def _arrays(samples):
    keep = [C.create_string_buffer(s, max(len(s), 1)) for s in samples]
    ptrs = (C.c_void_p * len(samples))(*[C.addressof(k) for k in keep])
    sizes = (C.c_size_t * len(samples))(*[len(s) for s in samples])
    return keep, ptrs, sizes


def train_dict(L, samples, cap):
    keep, ptrs, sizes = _arrays(samples)
    out = C.create_string_buffer(max(cap, 1))
    n = _bind(L).zxc_train_dict(ptrs, sizes, len(samples), out, cap)
    return (n, out.raw[:n]) if n >= 0 else (n, b"")


def train_huf(L, samples, d):
    keep, ptrs, sizes = _arrays(samples)
    out = C.create_string_buffer(128)
    db = C.create_string_buffer(d, len(d))
    rc = _bind(L).zxc_train_dict_huf(ptrs, sizes, len(samples), db, len(d), out)
    return rc, out.raw


def dict_train(L, samples):
    keep, ptrs, sizes = _arrays(samples)
    out = C.create_string_buffer(65535 + 16 + 128)
    n = _bind(L).zxc_dict_train(ptrs, sizes, len(samples), out, len(out))
    return (n, out.raw[:n]) if n >= 0 else (n, b"")


def _records(rng, n):
    return [b'{"id": %d, "user": "user%d", "status": "%s", "tags": ["alpha", "beta"], "score": %d}\n' %
            (i, rng.randrange(97), rng.choice([b"active", b"suspended", b"deleted"]), rng.randrange(1000)) for i in range(n)]


def _corpora():
    rng = random.Random(9)
    recs = _records(rng, 4000)
    words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randrange(3, 10))) for _ in range(300)]
    text = b" ".join(rng.choice(words) for _ in range(200000))
    return {
        "many small json samples": [b"".join(recs[i:i + 3]) for i in range(0, 3000, 3)],
        "one large text sample": [text],
        "mixed sizes": [text[:70000], b"".join(recs[:50]), b"", text[70000:70007], rng.randbytes(5000), b"".join(recs[50:2000])],
        "random bytes (no frequent pattern: the corpus' tail)": [rng.randbytes(3000), rng.randbytes(100)],
        "a run": [bytes(5000)],
        "barely a k-gram": [b"abcde"],
        "more than 2^19 k-grams (sampled counts)": [text * 3, b"".join(recs)],
    }


def test_train_dict_is_the_reference_s_dictionary_byte_for_byte(product, ref):
    mine, theirs = product.lib(), ref.lib
    for name, samples in _corpora().items():
        for cap in (65535, 16384, 1000, 64):
            want = train_dict(theirs, samples, cap)
            got = train_dict(mine, samples, cap)
            assert got == want, (name, cap, got[0], want[0])
            assert want[0] > 0
    # the refusals
    for samples, cap in (([b"abcd"], 100), ([b"ab", b"c"], 100), ([b"x" * 100], 65536)):
        assert train_dict(mine, samples, cap)[0] == train_dict(theirs, samples, cap)[0] < 0
    L = _bind(mine)
    assert L.zxc_train_dict(None, None, 0, None, 0) == -12 and L.zxc_dict_train(None, None, 0, None, 0) == -12
    assert L.zxc_train_dict_huf(None, None, 0, None, 0, None) == -12


def _check_table_with_the_reference(ref, samples, d, huf, min_gain):
    """the reference compresses 4 KiB blocks with (dictionary, shared table) and decodes them again; the table pays"""
    import oracle_py
    data = b"".join(samples)[:200000]
    sizes = {}
    for table in (None, huf):
        o = oracle_py.CompressOpts(level=6, block_size=4096, checksum_enabled=1)
        kd = C.create_string_buffer(d, len(d))
        o.dict, o.dict_size = C.cast(kd, C.c_void_p), len(d)
        if table:
            kh = C.create_string_buffer(table, 128)
            o.dict_huf = C.cast(kh, C.c_void_p)
        cap = ref.lib.zxc_compress_bound(len(data))
        dst = C.create_string_buffer(cap)
        n = ref.lib.zxc_compress(data, len(data), dst, cap, C.byref(o))
        assert n > 0, n
        do = oracle_py.DecompressOpts(checksum_enabled=1)
        do.dict, do.dict_size, do.dict_huf = o.dict, o.dict_size, o.dict_huf
        out = C.create_string_buffer(len(data))
        assert ref.lib.zxc_decompress(dst, n, out, len(data), C.byref(do)) == len(data) and out.raw == data
        sizes[bool(table)] = n
    assert sizes[True] <= sizes[False] * (1 - min_gain), sizes
    return sizes


def _huf_cases(L, ref):
    corp = _corpora()
    for name, min_gain in (("many small json samples", 0.02), ("one large text sample", 0.0)):
        samples = corp[name]
        n, d = train_dict(L, samples, 16384)
        assert n > 0
        rc, huf = train_huf(L, samples, d)
        assert rc == 0 and any(huf), name
        lens = [b & 15 for b in huf] + [b >> 4 for b in huf]
        assert max(lens) <= 8 and abs(sum(2.0 ** -x for x in lens if x) - 1.0) < 1e-9, "a complete prefix code of at most 8 bits"
        _check_table_with_the_reference(ref, samples, d, huf, min_gain)
    # one call: a .zxd both libraries load, carrying the trained content and a table; its id is the id of (content, table)
    samples = corp["many small json samples"]
    n, zxd = dict_train(L, samples)
    assert n > 16 + 128
    for lib_ in (L, ref.lib):
        B = _bind(lib_)
        content, size, hp, did = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_uint32()
        buf = C.create_string_buffer(zxd, len(zxd))
        assert B.zxc_dict_load(buf, len(zxd), C.byref(content), C.byref(size), C.byref(hp), C.byref(did)) == 0
        got = C.string_at(content.value, size.value)
        assert got == train_dict(ref.lib, samples, 65535)[1] and hp.value
        assert did.value == B.zxc_dict_id(content, size.value, hp) != 0
    # no literals left (a run): the all-zero table, like the reference (src/lib/zxc_dict.c:566-573)
    rc, huf = train_huf(L, [bytes(20000)], bytes(64))
    assert rc == 0 and not any(huf)


def test_train_dict_huf_and_dict_train_over_the_mock_device(mockdev, ref):
    _huf_cases(mockdev, ref)


@pytest.mark.gpu
def test_train_dict_huf_and_dict_train_on_the_device(product, ref):
    assert product.lib().zxc_mi355x_device_count() >= 1
    _huf_cases(product.lib(), ref)
    # and the product's own codec takes the trained pair: compress + decompress with (dictionary, table) on the device
    samples = _corpora()["many small json samples"]
    n, d = train_dict(product.lib(), samples, 16384)
    rc, huf = train_huf(product.lib(), samples, d)
    data = b"".join(samples)[:100000]
    arc = product.compress(data, level=6, block_size=4096, seekable=False, checksum=True, dict_=d, dict_huf=huf)
    assert product.decompress(arc, checksum=True, dict_=d, dict_huf=huf) == data
    plain = product.compress(data, level=6, block_size=4096, seekable=False, checksum=True)
    assert len(arc) < 0.8 * len(plain)


def test_train_dict_huf_fails_loudly_without_a_device(product):
    if product.lib().zxc_mi355x_device_count() > 0:
        pytest.skip("a HIP device is present")
    rc, _ = train_huf(product.lib(), [b"hello world " * 100], b"hello world ")
    assert rc == -100
    assert dict_train(product.lib(), [b"hello world " * 100])[0] == -100
