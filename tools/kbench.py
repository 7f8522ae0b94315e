#!/usr/bin/env python3
"""Kernel experiment harness (development tool, GPU box only): times
zxc_decode_blocks_kernel on single data classes / levels with optional ablation flags.
Usage: python tools/kbench.py [class ...]   classes: text exe source records chem image16 mixed zeros
Env: KB_MIB (per-class MiB, default 16), KB_REPL (replicas, default 16), KB_LEVELS ("3"), KB_DBG ("0,1,2,...")"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import zxc_amd
from zxc_amd import corpus
import oracle_py

def run(name, data, level, dbgs, R, bs=65536):
    ref = oracle_py.Ref()
    comp = ref.compress(data, level, bs, True, False)
    s = zxc_amd.Seekable(comp); nb = s.num_blocks; total = s.decompressed_size
    base = s.plan()
    cs = (len(comp) + 255) & ~255; osz = (total + 255) & ~255
    dev = torch.device("cuda", 0)
    d_comp = torch.empty(R * cs + 256, dtype=torch.uint8, device=dev)
    h = torch.frombuffer(bytearray(comp), dtype=torch.uint8)
    for r in range(R): d_comp[r*cs:r*cs+len(comp)].copy_(h)
    jobs = np.tile(base, R); rep = np.repeat(np.arange(R, dtype=np.uint64), nb)
    jobs["comp_off"] += rep * np.uint64(cs); jobs["out_off"] += rep * np.uint64(osz)
    d_jobs = torch.frombuffer(bytearray(jobs.tobytes()), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(R * osz + 256, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(jobs.size, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    L = zxc_amd.lib()
    res = []
    for dbg in dbgs:
        zxc_amd.api.set_debug(L, dbg)
        def step(): zxc_amd.decode_blocks_device(d_comp.data_ptr(), d_jobs.data_ptr(), jobs.size, d_out.data_ptr(), d_st.data_ptr(), bs, False, stream)
        step(); torch.cuda.synchronize()
        if dbg == 0:
            st = d_st.cpu().numpy()
            ok = (st == jobs["out_len"].astype(np.int32)).all() and bytes(d_out[:total].cpu().numpy()) == data
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): step()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        res.append((dbg, ms, R * total / ms / 1e6))
    zxc_amd.api.set_debug(L, 0)
    print(f"{name:8s} L{level} ratio {total/len(comp):5.2f} blocks {jobs.size:6d} ok={ok} | " +
          " | ".join(f"dbg{d}: {ms:7.2f} ms {g:7.1f} GB/s" for d, ms, g in res), flush=True)

if __name__ == "__main__":
    classes = sys.argv[1:] or ["text", "source", "exe", "mixed"]
    mib = int(os.environ.get("KB_MIB", "16")); R = int(os.environ.get("KB_REPL", "16"))
    levels = [int(x) for x in os.environ.get("KB_LEVELS", "3").split(",")]
    dbgs = [int(x) for x in os.environ.get("KB_DBG", "0").split(",")]
    for c in classes:
        if c == "mixed": data = corpus.synth_silesia(mib << 20, seed=0)
        elif c == "zeros": data = bytes(mib << 20)
        else: data = corpus._GEN[c](mib << 20, corpus._rng(0, 1)).tobytes()
        for lv in levels: run(c, data, lv, dbgs, R)
