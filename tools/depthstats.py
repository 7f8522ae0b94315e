#!/usr/bin/env python3
"""Match dependency depth per 64 KiB block of the bench corpus classes (design tool, CPU only): level of a match = 1 + the deepest
match its source overlaps. python tools/depthstats.py"""
import sys, os, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from zxc_amd import corpus
import oracle_py, seqstats
ref = oracle_py.Ref(); O = oracle_py.Oracle()
for cls, frac in seqstats.CLASSES:
    data = corpus._GEN[cls](2 << 20, corpus._rng(0, 1)).tobytes(); comp = ref.compress(data, 3, 65536, True, False)
    t = O.seek_table(comp); depths = []; hist = []
    for b in range(t["n_blocks"]):
        blk = comp[t["comp_offsets"][b]: t["comp_offsets"][b] + t["comp_sizes"][b]]
        s = seqstats.parse_block(blk)
        if s is None: continue
        ll, ml, off, nlit = s
        E = np.cumsum(ll + ml); M = E - ml
        dep = np.zeros(65536 + 4096, dtype=np.int32); dj = np.zeros(len(ml), dtype=np.int32)
        for j in range(len(ml)):
            qa = M[j] - off[j]; qb = min(M[j], qa + ml[j])
            d = 1 + (dep[qa:qb].max() if qb > qa else 0)
            dep[M[j]:E[j]] = d; dj[j] = d
        depths.append(dj.max()); hist.append(dj)
    h = np.concatenate(hist)
    print(f"{cls:10s} w {frac:.2f}: block max depth mean {np.mean(depths):.0f} p90 {np.percentile(depths,90):.0f} max {max(depths)} | per-match depth mean {h.mean():.1f} | frac depth1 {np.mean(h==1):.2f} <=2 {np.mean(h<=2):.2f} <=4 {np.mean(h<=4):.2f} <=8 {np.mean(h<=8):.2f}")
