#!/usr/bin/env python3
"""CPU model of the decode kernel's batch dependency rounds (design tool): parses GLO blocks of a
reference-compressed corpus class and reports rounds per 64-sequence batch with/without redirect."""
import sys, os, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from zxc_amd import corpus
import oracle_py
def varint(b, i):
    b0 = b[i]
    if b0 < 0x80: return b0, i + 1
    if b0 < 0xC0: return (b0 & 0x3F) | (b[i+1] << 6), i + 2
    return (b0 & 0x1F) | (b[i+1] << 5) | (b[i+2] << 13), i + 3
def seqs_of_block(blk):
    if blk[0] != 1: return None
    csz = struct.unpack_from("<I", blk, 3)[0]; d = blk[8:8+csz]
    nseq, nlit = struct.unpack_from("<II", d, 0); enc_lit, enc_tok, _, enc_off = d[8], d[9], d[10], d[11]
    if enc_tok: return None
    o = 12; lit_comp = nlit
    if enc_lit: lit_comp = struct.unpack_from("<I", d, 12)[0]; o += 4
    tok = o + lit_comp; offs = tok + nseq; ext = offs + nseq * (1 if enc_off else 2)
    out = []; e = ext
    for i in range(nseq):
        t = d[tok + i]; ll = t >> 4; ml = t & 15
        off = 1 + (d[offs + i] if enc_off else d[offs + 2*i] | (d[offs + 2*i + 1] << 8))
        if ll == 15: v, e = varint(d, e); ll += v
        if ml == 15: v, e = varint(d, e); ml += v
        out.append((ll, ml + 5, off))
    return out
def sim(seqs, TILE=2048, redirect=True):
    rounds = []; i = 0; p = 0
    while i < len(seqs):
        batch = []; pos = p
        while len(batch) < 64 and i + len(batch) < len(seqs):
            ll, ml, off = seqs[i + len(batch)]
            if pos + ll + ml - p > TILE and batch: break
            est = pos; M = est + ll; E = M + ml; batch.append((M, E, off, ml)); pos = E
        i += len(batch); depth = []; qsrcs = []
        for k, (M, E, off, ml) in enumerate(batch):
            qa = M - off; qb = min(M, qa + ml); deps = [j for j in range(k) if batch[j][1] > qa and batch[j][0] < qb] if qb > p else []
            dep = 0 if not deps else 1 + max(depth[j] for j in deps)
            if redirect and deps and len(deps) == 1 and off >= ml:
                j = deps[0]; jM, jE, jo, _ = batch[j]
                if qa >= jM and qb <= jE and qb - jM <= jo and depth[j] == 0: dep = 0
            depth.append(dep)
        rounds.append(1 + max(depth) if depth else 0); p = pos
    return np.array(rounds)
ref = oracle_py.Ref(); O = oracle_py.Oracle()
for cls in sys.argv[1:] or ["text", "source", "exe", "chem", "records", "image16", "catalogue"]:
    data = corpus._GEN[cls](2 << 20, corpus._rng(0, 1)).tobytes(); comp = ref.compress(data, 3, 65536, True, False)
    t = O.seek_table(comp); r0 = []; r1 = []; ns = 0
    for b in range(t["n_blocks"]):
        s = seqs_of_block(comp[t["comp_offsets"][b]: t["comp_offsets"][b] + t["comp_sizes"][b]])
        if not s: continue
        ns += len(s); r0.append(sim(s, redirect=False)); r1.append(sim(s, redirect=True))
    if not r0: print(cls, "no GLO blocks"); continue
    r0 = np.concatenate(r0); r1 = np.concatenate(r1)
    print(f"{cls:10s} seqs/blk {ns/t['n_blocks']:6.0f} batches {len(r0):5d} | rounds no-redirect mean {r0.mean():.2f} p90 {np.percentile(r0,90):.0f} max {r0.max()} | redirect mean {r1.mean():.2f} p90 {np.percentile(r1,90):.0f}")
