"""ctypes binding of libzxc_mi355x.so — same function names as the reference C API
(include/zxc_buffer.h, include/zxc_seekable.h) plus the device-resident entry points of
include/zxc_mi355x.h."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# zxc_dev_job_t (include/zxc_mi355x.h)
JOB_DTYPE = np.dtype([("comp_off", "<u8"), ("out_off", "<u8"), ("comp_size", "<u4"), ("out_len", "<u4")])


class ZxcError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = int(code)
        super().__init__(f"{what}: {error_name(self.code)} ({self.code})")


class _CompressOpts(C.Structure):  # include/zxc_opts.h
    _fields_ = [("n_threads", C.c_int), ("level", C.c_int), ("block_size", C.c_size_t),
                ("checksum_enabled", C.c_int), ("seekable", C.c_int), ("dict", C.c_void_p),
                ("dict_size", C.c_size_t), ("dict_huf", C.c_void_p), ("progress_cb", C.c_void_p),
                ("user_data", C.c_void_p)]


class _DecompressOpts(C.Structure):  # include/zxc_opts.h
    _fields_ = [("n_threads", C.c_int), ("checksum_enabled", C.c_int), ("dict", C.c_void_p),
                ("dict_size", C.c_size_t), ("dict_huf", C.c_void_p), ("progress_cb", C.c_void_p),
                ("user_data", C.c_void_p)]


_READ_AT = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64)


class _Reader(C.Structure):  # zxc_reader_t, include/zxc_seekable.h
    _fields_ = [("read_at", _READ_AT), ("ctx", C.c_void_p), ("size", C.c_uint64)]


def lib_path():
    # The product library, always — unless a development harness under tools/ asks for an A/B build on purpose: both
    # ZXC_TOOLS_AB=1 and ZXC_LIB_VARIANT=<file> must be set (a stray ZXC_LIB_VARIANT alone is ignored: experiment builds may
    # produce wrong output by design, zxc_amd/csrc/zxc_experiments.h).
    if os.environ.get("ZXC_TOOLS_AB") == "1" and os.environ.get("ZXC_LIB_VARIANT"):
        return os.path.join(_HERE, os.environ["ZXC_LIB_VARIANT"])
    return os.path.join(_HERE, "libzxc_mi355x.so")


def lib():
    """The product library. Fails loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ImportError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(p)
        L.zxc_error_name.restype = C.c_char_p
        L.zxc_error_name.argtypes = [C.c_int]
        L.zxc_decompress.restype = C.c_int64
        L.zxc_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(_DecompressOpts)]
        L.zxc_compress.restype = C.c_int64
        L.zxc_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(_CompressOpts)]
        L.zxc_mi355x_encode_slot_stride.restype = C.c_uint32
        L.zxc_mi355x_encode_slot_stride.argtypes = [C.c_uint32]
        L.zxc_mi355x_encode_blocks_device.restype = C.c_int
        L.zxc_mi355x_encode_blocks_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                                      C.c_void_p, C.c_void_p, C.c_void_p]
        L.zxc_get_decompressed_size.restype = C.c_uint64
        L.zxc_get_decompressed_size.argtypes = [C.c_char_p, C.c_size_t]
        L.zxc_compress_bound.restype = C.c_uint64
        L.zxc_compress_bound.argtypes = [C.c_size_t]
        L.zxc_seekable_open.restype = C.c_void_p
        L.zxc_seekable_open.argtypes = [C.c_char_p, C.c_size_t]
        L.zxc_seekable_free.argtypes = [C.c_void_p]
        L.zxc_seekable_open_reader.restype = C.c_void_p
        L.zxc_seekable_open_reader.argtypes = [C.POINTER(_Reader)]
        for f in ("zxc_seekable_get_num_blocks",):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        L.zxc_seekable_get_decompressed_size.restype = C.c_uint64
        L.zxc_seekable_get_decompressed_size.argtypes = [C.c_void_p]
        for f in ("zxc_seekable_get_block_comp_size", "zxc_seekable_get_block_decomp_size"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p, C.c_uint32]
        L.zxc_seekable_decompress_range.restype = C.c_int64
        L.zxc_seekable_decompress_range.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_size_t]
        L.zxc_seekable_decompress_range_mt.restype = C.c_int64
        L.zxc_seekable_decompress_range_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64,
                                                       C.c_size_t, C.c_int]
        L.zxc_seekable_set_dict.restype = C.c_int
        L.zxc_seekable_set_dict.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p]
        L.zxc_mi355x_device_count.restype = C.c_int
        L.zxc_mi355x_set_device.argtypes = [C.c_int]
        L.zxc_mi355x_plan_seekable.restype = C.c_int64
        L.zxc_mi355x_plan_seekable.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]
        L.zxc_mi355x_decode_blocks_device.restype = C.c_int
        L.zxc_mi355x_decode_blocks_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                                      C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def error_name(code):
    try:
        return lib().zxc_error_name(int(code)).decode()
    except Exception:  # library missing: still give a readable message
        return f"zxc error {code}"


def get_decompressed_size(comp: bytes) -> int:
    return int(lib().zxc_get_decompressed_size(comp, len(comp)))


def compress(data: bytes, level=3, block_size=65536, seekable=True, checksum=False, raise_on_error=True, dict_=None,
             dict_huf=None):
    """zxc_compress(): host buffer in, v8 archive out (blocks encoded on the GPU)."""
    o = _CompressOpts(level=level, block_size=block_size, seekable=int(seekable), checksum_enabled=int(checksum))
    if dict_:
        _keep = (C.create_string_buffer(dict_, len(dict_)), C.create_string_buffer(dict_huf, 128) if dict_huf else None)
        o.dict = C.cast(_keep[0], C.c_void_p)
        o.dict_size = len(dict_)
        o.dict_huf = C.cast(_keep[1], C.c_void_p) if dict_huf else None
    cap = int(lib().zxc_compress_bound(len(data)))
    out = C.create_string_buffer(max(cap, 64))
    rc = lib().zxc_compress(data, len(data), out, cap, C.byref(o))
    if rc < 0:
        if raise_on_error:
            raise ZxcError(rc, "zxc_compress")
        return rc, b""
    return out.raw[:rc] if raise_on_error else (rc, out.raw[:rc])


def decompress(comp: bytes, capacity=None, checksum=False, raise_on_error=True, dict_=None, dict_huf=None):
    """zxc_decompress(): whole frame, host buffers in, host buffer out (blocks decode on the GPU)."""
    cap = get_decompressed_size(comp) if capacity is None else int(capacity)
    out = C.create_string_buffer(max(cap, 1))
    o = _DecompressOpts(checksum_enabled=int(checksum))
    if dict_:
        _keep = (C.create_string_buffer(dict_, len(dict_)), C.create_string_buffer(dict_huf, 128) if dict_huf else None)
        o.dict = C.cast(_keep[0], C.c_void_p)
        o.dict_size = len(dict_)
        o.dict_huf = C.cast(_keep[1], C.c_void_p) if dict_huf else None
    rc = lib().zxc_decompress(comp, len(comp), out if cap else None, cap, C.byref(o))
    if rc < 0:
        if raise_on_error:
            raise ZxcError(rc, "zxc_decompress")
        return rc, b""
    return out.raw[:rc] if raise_on_error else (rc, out.raw[:rc])


class Seekable:
    """zxc_seekable handle (include/zxc_seekable.h). Keeps `comp` alive: the C handle borrows it."""

    def __init__(self, comp: bytes = None, reader=None, size=None):
        """`comp`: the whole archive in memory (zxc_seekable_open). `reader(offset, length) -> bytes` + `size`:
        zxc_seekable_open_reader — only header, seek table and footer are read at open, so an archive whose
        blocks live elsewhere (e.g. each GPU holding its own block range) can still be opened as ONE table."""
        self._comp = comp
        if reader is not None:
            def _cb(_ctx, dst, n, off):
                try:
                    b = reader(int(off), int(n))
                    if b is None or len(b) != n:
                        return -11  # ZXC_ERROR_IO
                    C.memmove(dst, b, n)
                    return n
                except Exception:
                    return -11
            self._cb = _READ_AT(_cb)
            self._rd = _Reader(self._cb, None, int(size))
            self._h = lib().zxc_seekable_open_reader(C.byref(self._rd))
        else:
            self._h = lib().zxc_seekable_open(comp, len(comp))
        if not self._h:
            raise ZxcError(-6, "zxc_seekable_open (not a seekable archive)")

    def close(self):
        if self._h:
            lib().zxc_seekable_free(self._h)
            self._h = None

    __del__ = close

    @property
    def num_blocks(self):
        return int(lib().zxc_seekable_get_num_blocks(self._h))

    @property
    def decompressed_size(self):
        return int(lib().zxc_seekable_get_decompressed_size(self._h))

    def block_comp_size(self, i):
        return int(lib().zxc_seekable_get_block_comp_size(self._h, i))

    def block_decomp_size(self, i):
        return int(lib().zxc_seekable_get_block_decomp_size(self._h, i))

    def set_dict(self, dict_, dict_huf=None):
        return int(lib().zxc_seekable_set_dict(self._h, dict_, len(dict_), dict_huf))

    def decompress_range(self, offset, length, n_threads=None, raise_on_error=True):
        out = C.create_string_buffer(max(length, 1))
        if n_threads is None:
            rc = lib().zxc_seekable_decompress_range(self._h, out, length, offset, length)
        else:
            rc = lib().zxc_seekable_decompress_range_mt(self._h, out, length, offset, length, n_threads)
        if rc < 0:
            if raise_on_error:
                raise ZxcError(rc, "zxc_seekable_decompress_range")
            return rc, b""
        return out.raw[:rc] if raise_on_error else (rc, out.raw[:rc])

    def plan(self, first=0, count=None, comp_rebase=0):
        """Job table (numpy structured array) for blocks [first, first+count)."""
        count = self.num_blocks - first if count is None else count
        jobs = np.zeros(count, dtype=JOB_DTYPE)
        rc = lib().zxc_mi355x_plan_seekable(self._h, first, count, comp_rebase, jobs.ctypes.data)
        if rc < 0:
            raise ZxcError(rc, "zxc_mi355x_plan_seekable")
        return jobs


def decode_blocks_device(d_comp, d_jobs, n_jobs, d_out, d_status, block_size, verify_trailer=False, stream=0):
    """zxc_mi355x_decode_blocks_device(): raw device pointers (ints), asynchronous on `stream`."""
    rc = lib().zxc_mi355x_decode_blocks_device(C.c_void_p(d_comp), C.c_void_p(d_jobs), n_jobs, C.c_void_p(d_out),
                                               C.c_void_p(d_status), block_size, int(verify_trailer),
                                               C.c_void_p(stream))
    if rc < 0:
        raise ZxcError(rc, "zxc_mi355x_decode_blocks_device")


# ---- FILE* callers (include/zxc_stream.h). ctypes has no FILE*, so the C library's fopen/fclose are used.
_LIBC = None


def _libc():
    global _LIBC
    if _LIBC is None:
        L = C.CDLL(None)
        L.fopen.restype = C.c_void_p
        L.fopen.argtypes = [C.c_char_p, C.c_char_p]
        L.fclose.argtypes = [C.c_void_p]
        _LIBC = L
    return _LIBC


class _File:
    def __init__(self, path, mode):
        self.fp = _libc().fopen(os.fsencode(path), mode.encode())
        if not self.fp:
            raise OSError(f"fopen({path!r}, {mode!r}) failed")

    def __enter__(self):
        return self.fp

    def __exit__(self, *a):
        _libc().fclose(self.fp)


def _bind_stream(L):
    L.zxc_stream_compress.restype = C.c_int64
    L.zxc_stream_compress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_CompressOpts)]
    L.zxc_stream_decompress.restype = C.c_int64
    L.zxc_stream_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_DecompressOpts)]
    L.zxc_stream_get_decompressed_size.restype = C.c_int64
    L.zxc_stream_get_decompressed_size.argtypes = [C.c_void_p]
    L.zxc_seekable_open_file.restype = C.c_void_p
    L.zxc_seekable_open_file.argtypes = [C.c_void_p]
    return L


def _set_dict(o, dict_, dict_huf):
    """Point an opts struct at a dictionary (content + optional 128-byte shared table); returns the buffers to keep alive."""
    if not dict_:
        return None
    keep = (C.create_string_buffer(dict_, len(dict_)), C.create_string_buffer(dict_huf, 128) if dict_huf else None)
    o.dict = C.cast(keep[0], C.c_void_p)
    o.dict_size = len(dict_)
    o.dict_huf = C.cast(keep[1], C.c_void_p) if dict_huf else None
    return keep


def stream_compress(src_path, dst_path, level=3, block_size=65536, seekable=True, checksum=False, library=None, dict_=None,
                    dict_huf=None):
    """zxc_stream_compress(): file in, archive out. Returns bytes written or a negative zxc_error_t."""
    L = _bind_stream(library or lib())
    o = _CompressOpts(level=level, block_size=block_size, seekable=int(seekable), checksum_enabled=int(checksum))
    _keep = _set_dict(o, dict_, dict_huf)
    with _File(src_path, "rb") as fi, _File(dst_path, "wb") as fo:
        return int(L.zxc_stream_compress(fi, fo, C.byref(o)))


def stream_decompress(src_path, dst_path, checksum=False, library=None, dict_=None, dict_huf=None):
    """zxc_stream_decompress(): archive in, file out (dst_path None = integrity check only)."""
    L = _bind_stream(library or lib())
    o = _DecompressOpts(checksum_enabled=int(checksum))
    _keep = _set_dict(o, dict_, dict_huf)
    with _File(src_path, "rb") as fi:
        if dst_path is None:
            return int(L.zxc_stream_decompress(fi, None, C.byref(o)))
        with _File(dst_path, "wb") as fo:
            return int(L.zxc_stream_decompress(fi, fo, C.byref(o)))


def stream_get_decompressed_size(path, library=None):
    L = _bind_stream(library or lib())
    with _File(path, "rb") as fi:
        return int(L.zxc_stream_get_decompressed_size(fi))


# ---- push streaming (include/zxc_pstream.h). The same prototypes bind the product and (in tests) the reference library.
class _InBuf(C.Structure):  # zxc_inbuf_t
    _fields_ = [("src", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


class _OutBuf(C.Structure):  # zxc_outbuf_t
    _fields_ = [("dst", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


def _bind_pstream(L):
    L.zxc_cstream_create.restype = C.c_void_p
    L.zxc_cstream_create.argtypes = [C.POINTER(_CompressOpts)]
    L.zxc_cstream_free.argtypes = [C.c_void_p]
    L.zxc_cstream_compress.restype = C.c_int64
    L.zxc_cstream_compress.argtypes = [C.c_void_p, C.POINTER(_OutBuf), C.POINTER(_InBuf)]
    L.zxc_cstream_end.restype = C.c_int64
    L.zxc_cstream_end.argtypes = [C.c_void_p, C.POINTER(_OutBuf)]
    L.zxc_dstream_create.restype = C.c_void_p
    L.zxc_dstream_create.argtypes = [C.POINTER(_DecompressOpts)]
    L.zxc_dstream_free.argtypes = [C.c_void_p]
    L.zxc_dstream_decompress.restype = C.c_int64
    L.zxc_dstream_decompress.argtypes = [C.c_void_p, C.POINTER(_OutBuf), C.POINTER(_InBuf)]
    L.zxc_dstream_finished.restype = C.c_int
    L.zxc_dstream_finished.argtypes = [C.c_void_p]
    for f in ("zxc_cstream_in_size", "zxc_cstream_out_size", "zxc_dstream_in_size", "zxc_dstream_out_size"):
        getattr(L, f).restype = C.c_size_t
        getattr(L, f).argtypes = [C.c_void_p]
    return L


def pstream_compress(data: bytes, in_chunk, out_chunk, level=3, block_size=0, checksum=False, library=None):
    """Feed `data` to a zxc_cstream in chunks of in_chunk bytes, drain through an out buffer of out_chunk bytes, finish with
    zxc_cstream_end (the loop of the reference's header example, include/zxc_pstream.h:26-49).
    -> (0, archive) or (negative zxc_error_t, bytes produced so far)."""
    L = _bind_pstream(library or lib())
    o = _CompressOpts(level=level, block_size=block_size, checksum_enabled=int(checksum))
    cs = L.zxc_cstream_create(C.byref(o))
    if not cs:
        return None, b""
    src = C.create_string_buffer(data, max(len(data), 1))
    base = C.addressof(src)
    out_chunk = min(out_chunk, len(data) + len(data) // 8 + (1 << 20))  # (no point in allocating more than the archive can take)
    obuf = C.create_string_buffer(max(out_chunk, 1))
    out = _OutBuf(C.addressof(obuf), out_chunk, 0)
    blob = bytearray()
    try:
        off = 0
        while off < len(data):
            n = min(in_chunk, len(data) - off)
            inb = _InBuf(base + off, n, 0)
            while inb.pos < inb.size:
                r = L.zxc_cstream_compress(cs, C.byref(out), C.byref(inb))
                if r < 0:
                    return int(r), bytes(blob)
                if out.pos:
                    blob += C.string_at(out.dst, out.pos)
                    out.pos = 0
                elif r > 0 and out.size == 0:
                    raise RuntimeError("no progress with an empty out buffer")
            off += n
        while True:
            r = L.zxc_cstream_end(cs, C.byref(out))
            if r < 0:
                return int(r), bytes(blob)
            if out.pos:
                blob += C.string_at(out.dst, out.pos)
                out.pos = 0
            if r == 0:
                return 0, bytes(blob)
    finally:
        L.zxc_cstream_free(cs)


def pstream_decompress(comp: bytes, in_chunk, out_chunk, checksum=False, library=None):
    """Feed `comp` to a zxc_dstream in chunks of in_chunk bytes, drain through an out buffer of out_chunk bytes, until the decoder
    neither consumes nor produces (reference tests/test_pstream_api.c:107-172).
    -> (rc of the last call, decoded bytes, finished flag, input bytes consumed)."""
    L = _bind_pstream(library or lib())
    o = _DecompressOpts(checksum_enabled=int(checksum))
    ds = L.zxc_dstream_create(C.byref(o))
    if not ds:
        return None, b"", 0, 0
    src = C.create_string_buffer(comp, max(len(comp), 1))
    base = C.addressof(src)
    hint = int.from_bytes(comp[-12:-4], "little") if len(comp) >= 12 else 0  # the footer's size: a hint only (mutants lie)
    out_chunk = min(out_chunk, max(hint, 1 << 16) + (4 << 20))
    obuf = C.create_string_buffer(max(out_chunk, 1))
    out = _OutBuf(C.addressof(obuf), out_chunk, 0)
    dec = bytearray()
    try:
        off = 0
        while True:
            n = min(in_chunk, len(comp) - off)
            inb = _InBuf(base + off if n else None, n, 0)
            r = L.zxc_dstream_decompress(ds, C.byref(out), C.byref(inb))
            if out.pos:
                dec += C.string_at(out.dst, out.pos)
                out.pos = 0
            off += inb.pos
            if r < 0:
                return int(r), bytes(dec), 0, off
            if (off >= len(comp) or L.zxc_dstream_finished(ds)) and inb.pos == 0 and r == 0:
                return 0, bytes(dec), int(L.zxc_dstream_finished(ds)), off
    finally:
        L.zxc_dstream_free(ds)


def set_debug(lib_handle, flags: int) -> None:
    """tools/ only: the kernel debug switches exist in libraries built with -DZXC_EXPERIMENT (tools/build_variant.sh <name>
    -DZXC_EXPERIMENT, selected with ZXC_LIB_VARIANT); the release library does not export the setter."""
    if not hasattr(lib_handle, "zxc_mi355x__set_debug"):
        if flags:
            raise RuntimeError("this library was built without -DZXC_EXPERIMENT: set ZXC_LIB_VARIANT to an experiment build")
        return
    lib_handle.zxc_mi355x__set_debug(flags)
