#!/usr/bin/env python3
"""Sequence statistics of the bench corpus (design tool, CPU only): per data class of one silesia-like tile, level 3 / 64 KiB blocks:
offset reach (which window size serves which share of the matches / of the match bytes), run-length histograms, sequences per
fixed output span. python tools/seqstats.py [tile_fraction]"""
import sys, os, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from zxc_amd import corpus
import oracle_py

def parse_block(blk):
    """-> (ll, ml, off) int arrays of a GLO block with raw sections, else None"""
    if blk[0] != 1: return None
    csz = struct.unpack_from("<I", blk, 3)[0]; d = np.frombuffer(blk, dtype=np.uint8, count=csz, offset=8)
    nseq, nlit = struct.unpack_from("<II", blk, 8); enc_lit, enc_tok, enc_off = blk[16], blk[17], blk[19]
    if enc_tok: return None
    o = 12; lit_comp = nlit
    if enc_lit: lit_comp = struct.unpack_from("<I", blk, 20)[0]; o += 4
    tok = o + lit_comp; offs = tok + nseq; ext = offs + nseq * (1 if enc_off else 2)
    t = d[tok:tok + nseq].astype(np.int64); ll = t >> 4; ml = t & 15
    if enc_off: off = 1 + d[offs:offs + nseq].astype(np.int64)
    else: off = 1 + d[offs:offs + 2 * nseq].view(np.uint16).astype(np.int64)
    # varints in stream order: LL escape before ML escape of the same sequence
    e = ext; db = bytes(d)
    escL = np.nonzero(ll == 15)[0]; escM = np.nonzero(ml == 15)[0]
    order = sorted([(i, 0) for i in escL] + [(i, 1) for i in escM])
    for i, w in order:
        b0 = db[e]
        if b0 < 0x80: v = b0; e += 1
        elif b0 < 0xC0: v = (b0 & 0x3F) | (db[e + 1] << 6); e += 2
        else: v = (b0 & 0x1F) | (db[e + 1] << 5) | (db[e + 2] << 13); e += 3
        if w == 0: ll[i] += v
        else: ml[i] += v
    return ll, ml + 5, off, nlit

def main():
    ref = oracle_py.Ref(); O = oracle_py.Oracle()
    n_per = int(float(sys.argv[1]) * (1 << 20)) if len(sys.argv) > 1 else (4 << 20)
    tot = {}
    for cls, frac in [(c, f) for c, f in CLASSES]:
        data = corpus._GEN[cls](n_per, corpus._rng(0, 1)).tobytes(); comp = ref.compress(data, 3, 65536, True, False)
        t = O.seek_table(comp)
        LL = []; ML = []; OFF = []; POS = []; nb = 0; nraw = 0; lits = 0
        for b in range(t["n_blocks"]):
            blk = comp[t["comp_offsets"][b]: t["comp_offsets"][b] + t["comp_sizes"][b]]
            s = parse_block(blk)
            if s is None: nraw += 1; continue
            ll, ml, off, nlit = s
            M = np.cumsum(ll + ml) - ml
            LL.append(ll); ML.append(ml); OFF.append(off); POS.append(M); nb += 1; lits += nlit
        if not nb: print(cls, "no GLO blocks"); continue
        ll = np.concatenate(LL); ml = np.concatenate(ML); off = np.concatenate(OFF)
        out = (ll + ml).sum()
        print(f"== {cls} (weight {frac:.2f}): GLO blocks {nb}, raw {nraw}, seq/block {ll.size/nb:.0f}, bytes/seq {out/ll.size:.1f}, lit share {ll.sum()/out:.3f}, comp ratio {len(data)/len(comp):.2f}")
        for W in (1024, 2048, 3072, 4096, 8192, 16384, 32768):
            near = off <= W
            print(f"   off<={W:5d}: {near.mean():.3f} of matches, {ml[near].sum()/ml.sum():.3f} of match bytes")
        print("   ml  <=8 %.3f <=12 %.3f <=16 %.3f <=20 %.3f <=32 %.3f <=64 %.3f <=128 %.3f | mean %.1f" % tuple([(ml <= x).mean() for x in (8, 12, 16, 20, 32, 64, 128)] + [ml.mean()]))
        print("   ll  ==0 %.3f <=4 %.3f <=8 %.3f <=12 %.3f <=16 %.3f <=32 %.3f <=64 %.3f | mean %.1f" % tuple([(ll == 0).mean()] + [(ll <= x).mean() for x in (4, 8, 12, 16, 32, 64)] + [ll.mean()]))
        print("   overlap (off<ml) %.3f, off<16 %.3f, off<4 %.3f" % ((off < ml).mean(), (off < 16).mean(), (off < 4).mean()))
        tot[cls] = (frac, ll, ml, off)
    # weighted mix
    print("== silesia-like mix (weighted by class share)")
    w = np.concatenate([np.full(v[1].size, v[0] / v[1].size) for v in tot.values()]); w /= w.sum()
    ll = np.concatenate([v[1] for v in tot.values()]); ml = np.concatenate([v[2] for v in tot.values()]); off = np.concatenate([v[3] for v in tot.values()])
    for W in (1024, 2048, 3072, 4096, 8192, 16384, 32768):
        near = off <= W
        print(f"   off<={W:5d}: {(w*near).sum():.3f} of matches, {(w*ml*near).sum()/(w*ml).sum():.3f} of match bytes")
    print("   ml  " + " ".join(f"<={x} {(w*(ml<=x)).sum():.3f}" for x in (8, 12, 16, 20, 32, 64, 128)))
    print("   ll  " + " ".join(f"<={x} {(w*(ll<=x)).sum():.3f}" for x in (0, 4, 8, 12, 16, 32, 64)))

_agg = {}
for _c, _f in corpus._SILESIA_MIX: _agg[_c] = _agg.get(_c, 0.0) + _f
CLASSES = sorted(_agg.items())
if __name__ == "__main__":
    main()
