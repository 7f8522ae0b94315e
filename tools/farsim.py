#!/usr/bin/env python3
"""Far-read simulator (design tool, CPU only): per corpus class, matches whose source is older than an LDS ring of argv[1] bytes
(default 4096), the distinct 64-byte sectors / 128-byte lines they touch per batch, and what an LRU of N sectors would catch.
python tools/farsim.py [ring_bytes]"""
import sys, numpy as np, collections
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from zxc_amd import corpus
import oracle_py, seqstats
ref = oracle_py.Ref(); O = oracle_py.Oracle()
RING = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tot = collections.Counter()
for cls, frac in seqstats.CLASSES:
    data = corpus._GEN[cls](2 << 20, corpus._rng(0, 1)).tobytes(); comp = ref.compress(data, 3, 65536, True, False)
    t = O.seek_table(comp)
    nfar = 0; nsect = 0; nline = 0; hits = {8: 0, 16: 0, 32: 0, 64: 0, 128: 0}; nm = 0; nblk = 0
    for b in range(t["n_blocks"]):
        blk = comp[t["comp_offsets"][b]: t["comp_offsets"][b] + t["comp_sizes"][b]]
        s = seqstats.parse_block(blk)
        if s is None: continue
        ll, ml, off, nlit = s; nblk += 1
        E = np.cumsum(ll + ml); M = E - ml
        caches = {n: collections.OrderedDict() for n in hits}
        i = 0; p = 0
        while i < len(ml):
            j = min(i + 64, len(ml))
            # batch [i, j): tile end
            while E[j - 1] - p > 3584 and j > i + 1: j -= 1
            z_new = (E[j - 1] + 15) & ~15; ring_lo = max(0, z_new - RING)
            sects = set(); 
            for q in range(i, j):
                nm += 1
                qa = M[q] - off[q]
                if qa < ring_lo and ml[q] <= 128:
                    nfar += 1
                    for sct in range(qa >> 6, ((qa + ml[q] - 1) >> 6) + 1): sects.add(sct)
            nsect += len(sects)
            nline += len(set(x >> 1 for x in sects))
            for sct in sects:
                for n, c in caches.items():
                    if sct in c: hits[n] += 1; c.move_to_end(sct)
                    else:
                        c[sct] = 1
                        if len(c) > n: c.popitem(last=False)
            p = E[j - 1]; i = j
    print(f"{cls:10s} w {frac:.2f}: far matches/block {nfar/nblk:7.1f} ({nfar/nm:.2f} of matches)  distinct sectors per batch-sum/block {nsect/nblk:7.1f}  128B lines {nline/nblk:7.1f}  LRU hit rate " + " ".join(f"{n}:{hits[n]/max(nsect,1):.2f}" for n in hits))
    tot["far"] += frac * nfar / nblk; tot["sect"] += frac * nsect / nblk; tot["line"] += frac * nline / nblk
    for n in hits: tot[n] += frac * hits[n] / nblk
print(f"mix: far/block {tot['far']:.0f} sectors/block {tot['sect']:.0f} lines/block {tot['line']:.0f} hits/block " + " ".join(f"{n}:{tot[n]:.0f}" for n in hits))
