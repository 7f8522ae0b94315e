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
    def __init__(self, variant=None):
        """variant="own": the build with the output-owner executor in the lean kernel (libzxc_wave_emu_own.so)."""
        build()
        variant = variant or os.environ.get("ZXC_EMU_VARIANT")
        L = self.lib = C.CDLL(SO.replace(".so", f"_{variant}.so") if variant else SO)
        L.emu_decode_blocks.restype = C.c_int
        L.emu_decode_blocks.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_uint32, C.c_int, C.c_char_p, C.c_uint32, C.c_char_p]

    def decode_jobs(self, comp: bytes, jobs: np.ndarray, out_bytes: int, block_size: int, verify_trailer=False,
                    dict_=None, dict_huf=None, cap_override=0, ck_apart=True):
        """Runs every job through one emulated wavefront. -> (status int32[n], output bytes).
        cap_override: the strict per-block capacity of zxc_decompress_block_safe (0: block_size + 2112)."""
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        self.lib.emu_set_cap_override(int(cap_override))
        self.lib.emu_set_ck_apart(int(ck_apart))  # (checksums by zxc_block_checksum_kernel beside the decode, or inside the decode kernels)
        out = C.create_string_buffer(max(out_bytes, 1))
        status = np.full(jobs.size, -999, dtype=np.int32)
        self.lib.emu_decode_blocks(comp, len(comp), jobs.ctypes.data, jobs.size, out, out_bytes, status.ctypes.data,
                                   block_size, int(verify_trailer), dict_, len(dict_) if dict_ else 0, dict_huf)
        self.lib.emu_set_cap_override(0)
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

    def encode(self, data: bytes, level=3, block_size=65536, checksum=False, seekable=True, dict_=None, dict_id=0) -> bytes:
        """The encode kernels on the emulator, one wavefront per block, assembled into a v8 archive the way
        zxc_compress (zxc_host.c) does: file header, blocks, EOF block, optional seek table, footer."""
        import oracle_py
        L = self.lib
        L.emu_encode_slot_stride.restype = C.c_uint32
        L.emu_encode_blocks.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint32]
        nb = (len(data) + block_size - 1) // block_size
        stride = L.emu_encode_slot_stride(block_size)
        slots = C.create_string_buffer(max(nb * stride, 1))
        sizes = np.zeros(max(nb, 1), dtype=np.uint32)
        if nb:
            L.emu_encode_blocks(data, len(data), block_size, level, int(checksum), slots, sizes.ctypes.data, dict_, len(dict_) if dict_ else 0)
        O = oracle_py.Oracle().lib
        O.zxo_hash16.argtypes = [C.c_char_p]
        O.zxo_hash8.argtypes = [C.c_char_p]
        hdr = bytearray(16)
        hdr[0:4] = (0x9CB02EF5).to_bytes(4, "little")
        hdr[4] = 8
        hdr[5] = block_size.bit_length() - 1
        hdr[6] = (0x80 if checksum else 0) | (0x40 if dict_ else 0)
        if dict_:
            hdr[7:11] = int(dict_id).to_bytes(4, "little")  # (docs/FORMAT.md §3: dict_id at byte 7 when HAS_DICTIONARY)
        hdr[14:16] = int(O.zxo_hash16(bytes(hdr))).to_bytes(2, "little")
        out = bytearray(hdr)
        gh = 0
        raw = slots.raw
        for b in range(nb):
            blk = raw[b * stride: b * stride + int(sizes[b])]
            out += blk
            if checksum:
                gh = (((gh << 1) | (gh >> 31)) & 0xFFFFFFFF) ^ int.from_bytes(blk[-4:], "little")
        eof = bytearray(8)
        eof[0] = 255
        eof[7] = O.zxo_hash8(bytes(eof))
        out += eof
        if seekable and nb:
            sek = bytearray(8)
            sek[0] = 254
            sek[3:7] = (4 * nb).to_bytes(4, "little")
            sek[7] = O.zxo_hash8(bytes(sek))
            out += sek + sizes[:nb].astype("<u4").tobytes()
        out += len(data).to_bytes(8, "little") + (gh if checksum else 0).to_bytes(4, "little")
        self.last_sizes = sizes[:nb].copy()
        return bytes(out)


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
