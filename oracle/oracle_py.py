"""ctypes loaders for the two CPU checkers — TEST INFRASTRUCTURE ONLY.

* ``Oracle``  — oracle/liboracle_zxc.so, our plain-C restatement (zxc_oracle.c).
* ``Ref``     — oracle/_ref/libzxc_ref.so, the unmodified reference compiled by
  oracle/Makefile (present when it was built in the container; travels to the
  GPU box as a prebuilt file).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module. The product package (zxc_amd) must never import it.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle_zxc.so")
REF_SO = os.path.join(HERE, "_ref", "libzxc_ref.so")


def build(quiet=True):
    """(Re)build the checkers; building the checker is not using it."""
    subprocess.run(["make", "-C", HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


class SeekTable(C.Structure):
    _fields_ = [("n_blocks", C.c_uint32), ("block_size", C.c_uint32), ("total_decomp", C.c_uint64),
                ("has_checksum", C.c_int), ("dict_id", C.c_uint32),
                ("comp_sizes", C.POINTER(C.c_uint32)), ("comp_offsets", C.POINTER(C.c_uint64))]


class BlockStats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("type", "n_sequences", "n_literals", "enc_lit", "enc_tok",
                                          "enc_off", "lit_bytes", "tok_bytes", "off_bytes",
                                          "extra_bytes", "n_varints")] + \
               [("match_bytes", C.c_uint64), ("off_hist", C.c_uint64 * 17), ("overlap_matches", C.c_uint32)]


class OracleCtx(C.Structure):
    _fields_ = [("block_size", C.c_uint32), ("checksum_enabled", C.c_int), ("dict", C.c_void_p),
                ("dict_size", C.c_size_t), ("dict_huf", C.c_void_p), ("strict_tail", C.c_int)]


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        L = self.lib = C.CDLL(ORACLE_SO)
        L.zxo_decompress.restype = C.c_int64
        L.zxo_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int,
                                     C.c_char_p, C.c_size_t, C.c_char_p]
        L.zxo_decode_block.restype = C.c_int
        L.zxo_decode_block.argtypes = [C.POINTER(OracleCtx), C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.zxo_seek_table_parse.restype = C.c_int
        L.zxo_seek_table_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(SeekTable)]
        L.zxo_seek_table_free.argtypes = [C.POINTER(SeekTable)]
        L.zxo_seekable_decompress_range.restype = C.c_int64
        L.zxo_seekable_decompress_range.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(SeekTable),
                                                    C.c_void_p, C.c_size_t, C.c_uint64, C.c_size_t]
        L.zxo_checksum32.restype = C.c_uint32
        L.zxo_checksum32.argtypes = [C.c_char_p, C.c_size_t]
        L.zxo_block_stats.restype = C.c_int
        L.zxo_block_stats.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(BlockStats)]
        L.zxo_hash8.restype = C.c_uint8
        L.zxo_hash16.restype = C.c_uint16
        L.zxo_huf_decode_section.restype = C.c_int
        L.zxo_huf_decode_section.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]

    def decompress(self, comp: bytes, cap: int, checksum=False, dict_=None, dict_huf=None):
        """-> (rc, bytes). rc >= 0 is the decoded size, < 0 a zxc_error_t."""
        out = C.create_string_buffer(max(cap, 1))
        rc = self.lib.zxo_decompress(comp, len(comp), out if cap else None, cap, int(checksum),
                                     dict_, len(dict_) if dict_ else 0, dict_huf)
        return rc, out.raw[:max(rc, 0)]

    def decode_block(self, blk: bytes, block_size: int, cap=None, checksum=False, strict_tail=False):
        """strict_tail: the reference's zxc_decompress_block_safe decoders (exact checks only)."""
        cap = block_size + 2112 if cap is None else cap
        ctx = OracleCtx(block_size, int(checksum), None, 0, None, int(strict_tail))
        out = C.create_string_buffer(cap + 1)
        rc = self.lib.zxo_decode_block(C.byref(ctx), blk, len(blk), out, cap)
        return rc, out.raw[:max(rc, 0)]

    def seek_table(self, comp: bytes):
        t = SeekTable()
        if self.lib.zxo_seek_table_parse(comp, len(comp), C.byref(t)) != 0:
            return None
        res = dict(n_blocks=t.n_blocks, block_size=t.block_size, total=t.total_decomp,
                   has_checksum=t.has_checksum, dict_id=t.dict_id,
                   comp_sizes=[t.comp_sizes[i] for i in range(t.n_blocks)],
                   comp_offsets=[t.comp_offsets[i] for i in range(t.n_blocks + 1)])
        self.lib.zxo_seek_table_free(C.byref(t))
        return res

    def seekable_range(self, comp: bytes, offset: int, length: int):
        t = SeekTable()
        if self.lib.zxo_seek_table_parse(comp, len(comp), C.byref(t)) != 0:
            return None, b""
        out = C.create_string_buffer(max(length, 1))
        rc = self.lib.zxo_seekable_decompress_range(comp, len(comp), C.byref(t), out, length, offset, length)
        self.lib.zxo_seek_table_free(C.byref(t))
        return rc, out.raw[:max(rc, 0)]

    def block_stats(self, blk: bytes):
        st = BlockStats()
        rc = self.lib.zxo_block_stats(blk, len(blk), C.byref(st))
        return rc, st


class CompressOpts(C.Structure):
    # include/zxc_opts.h:58-78 (reference)
    _fields_ = [("n_threads", C.c_int), ("level", C.c_int), ("block_size", C.c_size_t),
                ("checksum_enabled", C.c_int), ("seekable", C.c_int), ("dict", C.c_void_p),
                ("dict_size", C.c_size_t), ("dict_huf", C.c_void_p), ("progress_cb", C.c_void_p),
                ("user_data", C.c_void_p)]


class DecompressOpts(C.Structure):
    # include/zxc_opts.h:80-95 (reference)
    _fields_ = [("n_threads", C.c_int), ("checksum_enabled", C.c_int), ("dict", C.c_void_p),
                ("dict_size", C.c_size_t), ("dict_huf", C.c_void_p), ("progress_cb", C.c_void_p),
                ("user_data", C.c_void_p)]


def bind_zxc_api(L):
    """Attach the public zxc C API prototypes (same for the reference .so and
    for the product libzxc_mi355x.so — that is the drop-in point)."""
    L.zxc_compress_bound.restype = C.c_uint64
    L.zxc_compress_bound.argtypes = [C.c_size_t]
    L.zxc_compress.restype = C.c_int64
    L.zxc_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(CompressOpts)]
    L.zxc_decompress.restype = C.c_int64
    L.zxc_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(DecompressOpts)]
    L.zxc_get_decompressed_size.restype = C.c_uint64
    L.zxc_get_decompressed_size.argtypes = [C.c_char_p, C.c_size_t]
    L.zxc_seekable_open.restype = C.c_void_p
    L.zxc_seekable_open.argtypes = [C.c_char_p, C.c_size_t]
    L.zxc_seekable_free.argtypes = [C.c_void_p]
    L.zxc_seekable_get_num_blocks.restype = C.c_uint32
    L.zxc_seekable_get_num_blocks.argtypes = [C.c_void_p]
    L.zxc_seekable_get_decompressed_size.restype = C.c_uint64
    L.zxc_seekable_get_decompressed_size.argtypes = [C.c_void_p]
    L.zxc_seekable_get_block_comp_size.restype = C.c_uint32
    L.zxc_seekable_get_block_comp_size.argtypes = [C.c_void_p, C.c_uint32]
    L.zxc_seekable_get_block_decomp_size.restype = C.c_uint32
    L.zxc_seekable_get_block_decomp_size.argtypes = [C.c_void_p, C.c_uint32]
    L.zxc_seekable_decompress_range.restype = C.c_int64
    L.zxc_seekable_decompress_range.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_size_t]
    L.zxc_seekable_decompress_range_mt.restype = C.c_int64
    L.zxc_seekable_decompress_range_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64,
                                                   C.c_size_t, C.c_int]
    return L


class Ref:
    """The unmodified reference library (oracle/_ref/libzxc_ref.so)."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        self.lib = bind_zxc_api(C.CDLL(REF_SO))

    def compress(self, data: bytes, level=3, block_size=65536, seekable=True, checksum=False) -> bytes:
        o = CompressOpts(level=level, block_size=block_size, seekable=int(seekable),
                         checksum_enabled=int(checksum))
        cap = self.lib.zxc_compress_bound(len(data))
        dst = C.create_string_buffer(cap)
        n = self.lib.zxc_compress(data, len(data), dst, cap, C.byref(o))
        if n < 0:
            raise RuntimeError(f"reference zxc_compress failed: {n}")
        return dst.raw[:n]

    def decompress(self, comp: bytes, cap: int, checksum=False):
        o = DecompressOpts(checksum_enabled=int(checksum))
        out = C.create_string_buffer(max(cap, 1))
        rc = self.lib.zxc_decompress(comp, len(comp), out if cap else None, cap, C.byref(o))
        return rc, out.raw[:max(rc, 0)]

    def seekable_range_mt(self, comp: bytes, offset: int, length: int, threads: int, dst=None):
        s = self.lib.zxc_seekable_open(comp, len(comp))
        if not s:
            return None, b""
        out = dst if dst is not None else C.create_string_buffer(max(length, 1))
        rc = self.lib.zxc_seekable_decompress_range_mt(s, out, length, offset, length, threads)
        self.lib.zxc_seekable_free(s)
        return rc, (out.raw[:max(rc, 0)] if dst is None else b"")


def bind_block_api(L):
    """Block API + contexts (reference include/zxc_buffer.h:204-486): the same prototypes bind the reference
    .so and the product libzxc_mi355x.so."""
    L.zxc_get_dict_id.restype = C.c_uint32
    L.zxc_get_dict_id.argtypes = [C.c_char_p, C.c_size_t]
    for f in ("zxc_compress_block_bound", "zxc_decompress_block_bound"):
        getattr(L, f).restype = C.c_uint64
        getattr(L, f).argtypes = [C.c_size_t]
    L.zxc_create_cctx.restype = C.c_void_p
    L.zxc_create_cctx.argtypes = [C.POINTER(CompressOpts)]
    L.zxc_free_cctx.argtypes = [C.c_void_p]
    L.zxc_create_dctx.restype = C.c_void_p
    L.zxc_create_dctx.argtypes = []
    L.zxc_free_dctx.argtypes = [C.c_void_p]
    L.zxc_compress_block.restype = C.c_int64
    L.zxc_compress_block.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(CompressOpts)]
    L.zxc_compress_cctx.restype = C.c_int64
    L.zxc_compress_cctx.argtypes = L.zxc_compress_block.argtypes
    for f in ("zxc_decompress_block", "zxc_decompress_block_safe", "zxc_decompress_dctx"):
        getattr(L, f).restype = C.c_int64
        getattr(L, f).argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(DecompressOpts)]
    return L


class BlockApi:
    """Thin helper over either library's Block API (one cctx + one dctx, reused)."""

    def __init__(self, lib):
        self.L = bind_block_api(lib)
        self.c = self.L.zxc_create_cctx(None)
        self.d = self.L.zxc_create_dctx()
        assert self.c and self.d

    def close(self):
        self.L.zxc_free_cctx(self.c)
        self.L.zxc_free_dctx(self.d)
        self.c = self.d = None

    def compress_block(self, data, level=3, checksum=False, block_size=0):
        o = CompressOpts(level=level, checksum_enabled=int(checksum), block_size=block_size)
        cap = int(self.L.zxc_compress_block_bound(len(data))) or 64
        dst = C.create_string_buffer(cap)
        rc = self.L.zxc_compress_block(self.c, data, len(data), dst, cap, C.byref(o))
        return rc, dst.raw[:max(rc, 0)]

    def decompress_block(self, blk, cap, checksum=False, safe=False, dict_=None, dict_huf=None):
        o = DecompressOpts(checksum_enabled=int(checksum))
        keep = None
        if dict_:
            keep = (C.create_string_buffer(dict_, len(dict_)), C.create_string_buffer(dict_huf, 128) if dict_huf else None)
            o.dict = C.cast(keep[0], C.c_void_p)
            o.dict_size = len(dict_)
            o.dict_huf = C.cast(keep[1], C.c_void_p) if dict_huf else None
        out = C.create_string_buffer(max(cap, 1))
        fn = self.L.zxc_decompress_block_safe if safe else self.L.zxc_decompress_block
        rc = fn(self.d, blk, len(blk), out, cap, C.byref(o))
        return rc, out.raw[:max(rc, 0)]
