#!/usr/bin/env python3
"""Per-phase shader-clock split of the decode loop on the bench mix (needs tools/build_variant.sh phases -DEXP_PHASES)."""
import os, sys
os.environ["ZXC_LIB_VARIANT"] = os.environ.get("ZXC_LIB_VARIANT", "libzxc_phases.so")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, zxc_amd, bench
data, comp, prep = bench.build_workload(64 << 20, 3, 65536)
R = 32
s = zxc_amd.Seekable(comp); nb = s.num_blocks; total = s.decompressed_size
base = s.plan(); dev = torch.device("cuda", 0)
cs = (len(comp) + 255) & ~255; osz = (total + 255) & ~255
d_comp = torch.empty(R * cs + 256, dtype=torch.uint8, device=dev)
h = torch.frombuffer(bytearray(comp), dtype=torch.uint8)
for r in range(R): d_comp[r*cs:r*cs+len(comp)].copy_(h)
jobs = np.tile(base, R); rep = np.repeat(np.arange(R, dtype=np.uint64), nb)
jobs["comp_off"] += rep * np.uint64(cs); jobs["out_off"] += rep * np.uint64(osz)
d_jobs = torch.frombuffer(bytearray(jobs.tobytes()), dtype=torch.uint8).to(dev)
d_out = torch.zeros(R * osz + 256, dtype=torch.uint8, device=dev)
d_st = torch.zeros(jobs.size, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
def step(): zxc_amd.decode_blocks_device(d_comp.data_ptr(), d_jobs.data_ptr(), jobs.size, d_out.data_ptr(), d_st.data_ptr(), 65536, False, stream)
step(); step(); torch.cuda.synchronize()
out = d_out[:osz].cpu().numpy()  # replica 0
ph = np.stack([out[int(o):int(o) + 32].view(np.uint32) for o in base["out_off"]]).astype(np.float64)
ph = ph[(ph < 5e7).all(axis=1)]  # RAW blocks never reach the sequence loop: their bytes are data
names = ["parse+scan", "literals", "dep analysis", "round 0", "rounds 1+", "flush", "loop tail", "giant"]
tot = ph.sum()
print("share of wave time per phase (all blocks):")
for i, n in enumerate(names): print(f"  {n:14s} {100 * ph[:, i].sum() / tot:5.1f} %   mean {ph[:, i].mean():9.0f} clk/block")
print(f"  total mean {ph.sum(axis=1).mean():.0f} clk/block")
