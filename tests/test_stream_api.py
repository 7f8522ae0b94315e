"""FILE* callers (include/zxc_stream.h): the same batches of blocks as the buffer API,
fed from / drained to files. Differential against the unmodified reference's own zxc_stream_* where built."""
import ctypes as C
import hashlib
import os

import pytest

from conftest import GOLDEN, read


def test_stream_size_probe_needs_no_gpu(product, manifest, tmp_path):
    for name, meta in list(manifest["synth"].items())[:6]:
        p = os.path.join(GOLDEN, "synth", name + ".zxc")
        assert product.api.stream_get_decompressed_size(p) == meta["size"]
    bad = tmp_path / "bad.zxc"
    bad.write_bytes(b"\0" * 64)
    assert product.api.stream_get_decompressed_size(str(bad)) == -4  # BAD_MAGIC
    tiny = tmp_path / "tiny.zxc"
    tiny.write_bytes(b"abc")
    assert product.api.stream_get_decompressed_size(str(tiny)) == -3  # SRC_TOO_SMALL


def test_stream_fails_loudly_without_gpu(product, tmp_path):
    """No CPU codec behind the FILE* callers either: without a HIP device they return GPU_UNAVAILABLE (-100)."""
    if product.lib().zxc_mi355x_device_count() > 0:
        pytest.skip("a GPU is present")
    p = os.path.join(GOLDEN, "synth", "lorem_100k_l3_b64k.zxc")
    assert product.api.stream_decompress(p, str(tmp_path / "o.bin")) == -100
    src = tmp_path / "in.bin"
    src.write_bytes(b"hello world " * 1000)
    assert product.api.stream_compress(str(src), str(tmp_path / "a.zxc")) == -100


@pytest.mark.gpu
def test_stream_decompress_matches_inputs(product, manifest, synth_inputs, tmp_path, monkeypatch):
    monkeypatch.setenv("ZXC_STREAM_BATCH_BYTES", str(256 << 10))  # several launches per file
    out = tmp_path / "out.bin"
    for name, meta in manifest["synth"].items():
        p = os.path.join(GOLDEN, "synth", name + ".zxc")
        rc = product.api.stream_decompress(p, str(out), checksum=bool(meta["checksum"]))
        assert rc == meta["size"], (name, rc)
        assert out.read_bytes() == synth_inputs[meta["input"]], name
        assert product.api.stream_decompress(p, None) == meta["size"]  # integrity-only mode


@pytest.mark.gpu
def test_stream_errors_match_buffer_api(product, manifest, tmp_path):
    import random
    rng = random.Random(5)
    comp = bytearray(read("synth/mixed_384k_l3_b64k.zxc"))
    size = manifest["synth"]["mixed_384k_l3_b64k"]["size"]
    f = tmp_path / "m.zxc"
    for _ in range(12):
        m = bytearray(comp)
        m[rng.randrange(24, len(m) - 40)] ^= 1 << rng.randrange(8)
        f.write_bytes(bytes(m))
        a, _ = product.decompress(bytes(m), size, raise_on_error=False)
        b = product.api.stream_decompress(str(f), str(tmp_path / "o.bin"))
        assert (a < 0) == (b < 0), (a, b)
        if a >= 0:
            assert a == b
    f.write_bytes(bytes(comp[:len(comp) // 2]))  # truncated file
    assert product.api.stream_decompress(str(f), None) < 0


@pytest.mark.gpu
def test_stream_compress_round_trips_through_the_reference(product, ref, tmp_path, monkeypatch):
    from zxc_amd import corpus
    monkeypatch.setenv("ZXC_STREAM_BATCH_BYTES", str(2 << 20))  # 512 KiB of source per launch
    data = corpus.synth_silesia(3 << 20, seed=4) + b"tail"
    src = tmp_path / "in.bin"
    src.write_bytes(data)
    for level, seekable, ck in ((1, False, False), (3, True, False), (5, True, True)):
        arc = tmp_path / f"a{level}.zxc"
        n = product.api.stream_compress(str(src), str(arc), level=level, seekable=seekable, checksum=ck)
        assert n == arc.stat().st_size
        comp = arc.read_bytes()
        rc, out = ref.decompress(comp, len(data), checksum=ck)  # unmodified reference decoder
        assert rc == len(data) and out == data, (level, rc)
        back = tmp_path / "back.bin"
        assert product.api.stream_decompress(str(arc), str(back), checksum=ck) == len(data)
        assert hashlib.sha256(back.read_bytes()).digest() == hashlib.sha256(data).digest()
        # and the reference's own FILE* reader takes it too
        assert product.api.stream_decompress(str(arc), str(back), checksum=ck, library=ref.lib) == len(data)
        assert back.read_bytes() == data
    empty = tmp_path / "empty.bin"
    empty.write_bytes(b"")
    arc = tmp_path / "e.zxc"
    assert product.api.stream_compress(str(empty), str(arc)) == arc.stat().st_size
    assert ref.decompress(arc.read_bytes(), 0)[0] == 0


@pytest.mark.gpu
def test_stream_compress_with_dictionary(product, ref, tmp_path, monkeypatch):
    """zxc_stream_compress with opts.dict (reference: the same opts reach zxc_stream_engine_run, src/lib/zxc_driver.c:1038):
    the archive carries the dictionary id, the unmodified reference decodes it with the dictionary and refuses it without."""
    monkeypatch.setenv("ZXC_STREAM_BATCH_BYTES", str(2 << 20))
    rec = b"".join(b'{"id": %d, "name": "user%d", "status": "active", "tags": ["a", "b"]}\n' % (i, i % 97) for i in range(30000))
    d = rec[:8192]
    src = tmp_path / "in.bin"
    src.write_bytes(rec)
    arc = tmp_path / "d.zxc"
    plain = tmp_path / "p.zxc"
    n = product.api.stream_compress(str(src), str(arc), level=3, block_size=4096, checksum=True, dict_=d)
    assert n == arc.stat().st_size
    assert product.api.stream_compress(str(src), str(plain), level=3, block_size=4096, checksum=True) > n  # the dictionary pays
    import ctypes as C
    import oracle_py
    comp = arc.read_bytes()
    assert oracle_py.bind_block_api(ref.lib).zxc_get_dict_id(comp, len(comp)) != 0
    o = oracle_py.DecompressOpts(checksum_enabled=1)
    keep = C.create_string_buffer(d, len(d))
    o.dict, o.dict_size = C.cast(keep, C.c_void_p), len(d)
    out = C.create_string_buffer(len(rec))
    assert ref.lib.zxc_decompress(comp, len(comp), out, len(rec), C.byref(o)) == len(rec) and out.raw == rec
    assert ref.decompress(comp, len(rec), checksum=True)[0] == -15  # ZXC_ERROR_DICT_REQUIRED
    back = tmp_path / "back.bin"
    assert product.api.stream_decompress(str(arc), str(back), checksum=True, dict_=d) == len(rec)
    assert back.read_bytes() == rec
    assert product.api.stream_decompress(str(arc), str(back), checksum=True) == -15


@pytest.mark.gpu
def test_seekable_open_file(product, manifest, synth_inputs):
    L = product.api._bind_stream(product.lib())
    name = "text_200k_l3_b4k"
    data = synth_inputs[manifest["synth"][name]["input"]]
    with product.api._File(os.path.join(GOLDEN, "synth", name + ".zxc"), "rb") as fp:
        h = L.zxc_seekable_open_file(fp)
        assert h
        L.zxc_seekable_decompress_range.restype = C.c_int64
        buf = C.create_string_buffer(5000)
        rc = L.zxc_seekable_decompress_range(C.c_void_p(h), buf, 5000, C.c_uint64(12345), 5000)
        assert rc == 5000 and buf.raw == data[12345:17345]
        L.zxc_seekable_free(C.c_void_p(h))
