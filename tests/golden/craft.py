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


def ghi_block(oracle, seqs, literals):
    """One GHI block (type 2). seqs: [(ll, ml, off)], ml >= 5, 1 <= off <= 65536 (docs/FORMAT.md §5.3: 32-bit sequence words
    LL << 24 | (ML - 5) << 16 | (off - 1), 255 = varint-extended)."""
    words = bytearray(); ext = bytearray()
    for ll, ml, off in seqs:
        m = ml - 5
        words += struct.pack("<I", (min(ll, 255) << 24) | (min(m, 255) << 16) | (off - 1))
        if ll >= 255:
            ext += _varint(ll - 255)
        if m >= 255:
            ext += _varint(m - 255)
    body = bytes(literals) + bytes(words) + bytes(ext)
    pad = max(0, 32 - (len(words) + len(ext)))
    payload = struct.pack("<IIBBBB", len(seqs), len(literals), 0, 0, 0, 0) + body + bytes(pad)
    hdr = bytearray(8)
    hdr[0] = 2
    hdr[3:7] = struct.pack("<I", len(payload))
    oracle.lib.zxo_hash8.argtypes = [C.c_char_p]
    hdr[7] = oracle.lib.zxo_hash8(bytes(hdr))
    return bytes(hdr) + payload


def overflow_margin_frames(oracle):
    """Frames around the reference's 4x-batch output reserve (src/lib/zxc_decompress.c:626-656, :701-727, loops :1050-1103 /
    :1296-1384): one 4 KiB-block frame each (per-block capacity 4096 + 2112 = 6208), the interesting sequence varint-extended.
    Round 3 pinned the first three as the one allowed divergence (the exact decoders answered 6140 / -8 / -9 where the
    reference answers OVERFLOW); since round 4 the oracle and the kernels reproduce the reserve, and every decoder must give
    the reference's answer. -> {name: (frame bytes, the reference's answer through zxc_decompress)}"""
    lits = bytes((i * 7 + 3) & 255 for i in range(6200))
    good = [(6100, 5, 1)] + [(0, 5, 1)] * 7
    out = {}
    # 6100 + 5 + 3 x 33 + 32 > 6208: OVERFLOW, although the block decodes to 6140 bytes
    out["fits_exactly"] = (frame(oracle, [glo_block(oracle, good, lits[:6100])], 12, 6140), -10)
    out["size_mismatch"] = (frame(oracle, [glo_block(oracle, good, lits[:6100])], 12, 6000), -10)
    bad = [(6100, 5, 1)] + [(0, 5, 1)] * 3 + [(0, 5, 65000)] + [(0, 5, 1)] * 3
    out["bad_offset_behind_it"] = (frame(oracle, [glo_block(oracle, bad, lits[:6100])], 12, 6140), -10)
    # the same run inside the reserve: 6000 + 5 + 99 + 32 <= 6208
    out["inside_the_reserve"] = (frame(oracle, [glo_block(oracle, [(6000, 5, 1)] + [(0, 5, 1)] * 7, lits[:6000])], 12, 6040), 6040)
    # the long run in the last, incomplete group of four (1x loops: exact checks only)
    tail = [(1, 5, 1)] + [(0, 5, 1)] * 3 + [(6090, 5, 1), (0, 5, 1), (0, 5, 1)]
    out["tail_group_is_exact"] = (frame(oracle, [glo_block(oracle, tail, lits[:6091])], 12, 6126), 6126)
    # second sequence of its batch (two remain behind it): 6110 + 5 + 2 x 33 + 32 = 6213 > 6208 - 6 (decodes to 6151 bytes)
    second = [(1, 5, 1), (6110, 5, 1)] + [(0, 5, 1)] * 6
    out["second_of_its_batch"] = (frame(oracle, [glo_block(oracle, second, lits[:6111])], 12, 6151), -10)
    # a varint-extended MATCH: 1 + 6100 + 99 + 32 > 6208 (60 literals: a block with fewer than 57 has no 4x batch at all)
    out["match_escape"] = (frame(oracle, [glo_block(oracle, [(1, 6100, 1)] + [(0, 5, 1)] * 7, lits[:60])], 12, 6195), -10)
    out["match_escape_few_literals"] = (frame(oracle, [glo_block(oracle, [(1, 6100, 1)] + [(0, 5, 1)] * 7, lits[:1])], 12, 6136), 6136)
    # the literal reserve: 20 + (14 + 14 + 14) > 61 literals -> OVERFLOW at once, in front of the bad offset an exact decoder meets first
    litres = [(20, 5, 1), (14, 5, 60000), (14, 5, 1), (14, 5, 1)]
    out["literal_reserve"] = (frame(oracle, [glo_block(oracle, litres, lits[:61])], 12, 82), -10)
    # GHI: 4700 + 5 + 3 x 513 + 32 > 6208, the block itself decodes to 5840 bytes; and one inside the reserve
    g = [(4700, 5, 1)] + [(0, 5, 1)] * 7
    out["ghi_run"] = (frame(oracle, [ghi_block(oracle, g, lits[:5800])], 12, 5840), -10)
    g = [(3000, 5, 1)] + [(0, 5, 1)] * 7
    out["ghi_inside_the_reserve"] = (frame(oracle, [ghi_block(oracle, g, lits[:4100])], 12, 4140), 4140)
    return out


def random_reserve_blocks(oracle, seed, count):
    """Random single-block 4 KiB frames whose decoded size lands around the per-block capacity (6208) or whose literal stream
    runs out around the last sequences, built from many short sequences with varint-extended ones sprinkled in and the odd
    multi-KiB sequence (cut batches: the kernels' batch starts then fall inside the reference's groups of four). For the
    differential tests of the 4x-batch reserve: -> [(frame bytes, block bytes)]."""
    import random
    rng = random.Random(seed)
    out = []
    for _ in range(count):
        ghi = rng.random() < 0.35
        esc = 255 if ghi else 15
        target = rng.choice((rng.randrange(5800, 6500), rng.randrange(5800, 6500), rng.randrange(200, 6500)))
        seqs, produced, lits_used = [], 0, 0
        big_at = rng.randrange(0, 40) if rng.random() < 0.6 else -1
        while produced < target and len(seqs) < 900:
            r = rng.random()
            if len(seqs) == big_at:
                ll, ml = (rng.randrange(1500, 4200), 5) if rng.random() < 0.5 else (1, rng.randrange(1500, 4200))
            elif r < 0.08:
                ll, ml = esc + rng.randrange(0, 200), 5 + rng.randrange(0, 12)
            elif r < 0.16:
                ll, ml = rng.randrange(0, 12), 5 + (esc if ghi else 15) + rng.randrange(0, 200)
            else:
                ll, ml = rng.randrange(0, 15), 5 + rng.randrange(0, 15)
            if not seqs and ll == 0:
                ll = 1
            off = 1 if rng.random() < 0.97 else rng.randrange(1, 70000 if rng.random() < 0.3 else max(2, produced + ll))
            off = min(off, 65536)
            seqs.append((ll, ml, off))
            produced += ll + ml
            lits_used += ll
        # literal stream: exactly what the sequences use, a little more, or a little less (overrun near the end)
        n_lit = max(0, lits_used + rng.choice((0, 0, 5, 70, -3, -20, 1100)))
        lits = bytes((i * 13 + 5) & 255 for i in range(n_lit))
        blk = ghi_block(oracle, seqs, lits) if ghi else glo_block(oracle, seqs, lits, off8=False)
        out.append((frame(oracle, [blk], 12, min(produced + max(0, n_lit - lits_used), 1 << 20)), blk))
    return out
