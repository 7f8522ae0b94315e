"""Hand-built ZXC v8 frames for the tests (test infrastructure): GLO blocks from explicit (ll, ml, off) sequences, framed like
docs/FORMAT.md §3-§9 (file header 16 B with its CRC16, block header 8 B with its CRC8, EOF block, footer). The header hashes
come from the oracle library (zxo_hash16 / zxo_hash8), the same helpers tests/wave_emu/emu_py.py uses."""
import ctypes as C
import struct


def _varint(v):
    if v < 0x80:
        return bytes([v])
    if v < (1 << 14):
        return bytes([0x80 | (v & 0x3F), v >> 6])
    assert v < (1 << 21)
    return bytes([0xC0 | (v & 0x1F), (v >> 5) & 0xFF, v >> 13])


def glo_block(oracle, seqs, literals, off8=False):
    """One GLO block (type 1) with raw sections. seqs: [(ll, ml, off)], ml >= 5, off >= 1; literals: all literal bytes
    (those the sequences consume, then the trailing ones)."""
    tok = bytearray(); offs = bytearray(); ext = bytearray()
    for ll, ml, off in seqs:
        m = ml - 5
        tok.append((min(ll, 15) << 4) | min(m, 15))
        offs += bytes([off - 1]) if off8 else struct.pack("<H", off - 1)
        if ll >= 15:
            ext += _varint(ll - 15)
        if m >= 15:
            ext += _varint(m - 15)
    body = bytes(literals) + bytes(tok) + bytes(offs) + bytes(ext)
    pad = max(0, 32 - (len(tok) + len(offs) + len(ext)))   # FORMAT: at least 32 bytes behind the literal section
    payload = struct.pack("<IIBBBB", len(seqs), len(literals), 0, 0, 0, 1 if off8 else 0) + body + bytes(pad)
    hdr = bytearray(8)
    hdr[0] = 1
    hdr[3:7] = struct.pack("<I", len(payload))
    oracle.lib.zxo_hash8.argtypes = [C.c_char_p]
    hdr[7] = oracle.lib.zxo_hash8(bytes(hdr))
    return bytes(hdr) + payload


def frame(oracle, blocks, block_size_log2, total):
    """File header + blocks + EOF block + footer (no checksums, no seek table)."""
    hdr = bytearray(16)
    hdr[0:4] = (0x9CB02EF5).to_bytes(4, "little")
    hdr[4] = 8
    hdr[5] = block_size_log2
    oracle.lib.zxo_hash16.argtypes = [C.c_char_p]
    oracle.lib.zxo_hash8.argtypes = [C.c_char_p]
    hdr[14:16] = int(oracle.lib.zxo_hash16(bytes(hdr))).to_bytes(2, "little")
    eof = bytearray(8)
    eof[0] = 255
    eof[7] = oracle.lib.zxo_hash8(bytes(eof))
    return bytes(hdr) + b"".join(blocks) + bytes(eof) + int(total).to_bytes(8, "little") + (0).to_bytes(4, "little")


def overflow_margin_frames(oracle):
    """The three frames that exercise tests/test_oracle_golden.py:ALLOWED_DIVERGENCE: one 4 KiB-block frame whose only block
    opens with a 6100-byte literal run (a varint-extended sequence inside the reference's first 4x batch) and decodes to
    6140 bytes, i.e. 68 bytes short of the per-block capacity 4096 + 2112. The reference's batch check reserves the
    inline maxima of the three sequences behind it plus its wild-copy pad (src/lib/zxc_decompress.c:626-656: 6100 + 5 +
    3 x 33 + 32 > 6208) and answers OVERFLOW; an exact decoder fits the block and moves on to its real fate.
    -> {name: (frame bytes, what the exact decoders answer)}"""
    lits = bytes((i * 7 + 3) & 255 for i in range(6100))
    good = [(6100, 5, 1)] + [(0, 5, 1)] * 7
    out = {}
    out["fits_exactly"] = (frame(oracle, [glo_block(oracle, good, lits)], 12, 6140), 6140)          # ("ok", -10)
    out["size_mismatch"] = (frame(oracle, [glo_block(oracle, good, lits)], 12, 6000), -8)           # (-8, -10)
    bad = [(6100, 5, 1)] + [(0, 5, 1)] * 3 + [(0, 5, 65000)] + [(0, 5, 1)] * 3
    out["bad_offset_behind_it"] = (frame(oracle, [glo_block(oracle, bad, lits)], 12, 6140), -9)     # (-9, -10)
    return out
