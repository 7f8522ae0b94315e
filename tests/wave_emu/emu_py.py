"""ctypes loader of tests/wave_emu/libzxc_wave_emu.so (the decode kernels compiled for the CPU wave
emulator). TEST INFRASTRUCTURE ONLY: nothing in zxc_amd/ or bench.py's timed path may import it."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libzxc_wave_emu.so")
JOB_DTYPE = np.dtype([("comp_off", "<u8"), ("out_off", "<u8"), ("comp_size", "<u4"), ("out_len", "<u4")])


def build(quiet=True):
    subprocess.run(["make", "-C", HERE, "all"], check=True, stdout=subprocess.DEVNULL if quiet else None)


class Emu:
    def __init__(self):
        build()
        L = self.lib = C.CDLL(SO)
        L.emu_decode_blocks.restype = C.c_int
        L.emu_decode_blocks.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_uint32, C.c_int, C.c_char_p, C.c_uint32, C.c_char_p]

    def decode_jobs(self, comp: bytes, jobs: np.ndarray, out_bytes: int, block_size: int, verify_trailer=False,
                    dict_=None, dict_huf=None):
        """Runs every job through one emulated wavefront. -> (status int32[n], output bytes)."""
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        out = C.create_string_buffer(max(out_bytes, 1))
        status = np.full(jobs.size, -999, dtype=np.int32)
        self.lib.emu_decode_blocks(comp, len(comp), jobs.ctypes.data, jobs.size, out, out_bytes, status.ctypes.data,
                                   block_size, int(verify_trailer), dict_, len(dict_) if dict_ else 0, dict_huf)
        return status, out.raw[:out_bytes]

    def decode_seekable(self, comp: bytes, table: dict, **kw):
        """table = Oracle.seek_table(comp). Decodes every block at i * block_size (the seekable layout)."""
        n, bs, total = table["n_blocks"], table["block_size"], table["total"]
        jobs = np.zeros(n, dtype=JOB_DTYPE)
        jobs["comp_off"] = table["comp_offsets"][:n]
        jobs["comp_size"] = table["comp_sizes"]
        jobs["out_off"] = np.arange(n, dtype=np.uint64) * np.uint64(bs)
        jobs["out_len"] = np.minimum(bs, total - np.arange(n, dtype=np.int64) * bs).astype(np.uint32)
        return jobs, *self.decode_jobs(comp, jobs, total, bs, verify_trailer=bool(table["has_checksum"]) and kw.pop("verify", False), **kw)


def frame_jobs(comp: bytes):
    """Walks the block headers of a (seekable or plain) archive the way the host API does:
    -> (jobs for the data blocks, block_size, has_checksum, decoded size from the footer)."""
    bs = 1 << comp[5]
    ck = bool(comp[6] & 0x80)
    pos = 16
    rows = []
    while True:
        t = comp[pos]
        csz = int.from_bytes(comp[pos + 3:pos + 7], "little")
        if t == 255:
            break
        phys = 8 + csz + (4 if ck else 0)
        rows.append((pos, phys))
        pos += phys
    total = int.from_bytes(comp[-12:-4], "little")
    jobs = np.zeros(len(rows), dtype=JOB_DTYPE)
    for i, (o, n) in enumerate(rows):
        jobs[i] = (o, i * bs, n, max(0, min(bs, total - i * bs)))
    return jobs, bs, ck, total
