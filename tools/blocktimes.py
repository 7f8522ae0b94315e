#!/usr/bin/env python3
"""Per-block wall time inside one launch of the bench workload (needs the EXP_TIMES build:
tools/build_variant.sh times -DEXP_TIMES). Prints the duration distribution and the residency timeline."""
import os, sys
os.environ["ZXC_LIB_VARIANT"] = os.environ.get("ZXC_LIB_VARIANT", "libzxc_times.so"); os.environ["ZXC_TOOLS_AB"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, zxc_amd, bench
base_mib = int(os.environ.get("BT_MIB", "64")); R = int(os.environ.get("BT_REPL", "32"))
data, comp, prep = bench.build_workload(base_mib << 20, 3, 65536)
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
st = d_st.cpu().numpy().view(np.uint32)
start = (st >> 16).astype(np.int64); dur = (st & 0xFFFF).astype(np.float64) * 0.32  # us
piv = int(np.median(start))
start = (((start - piv + 0x8000) & 0xFFFF) - 0x8000).astype(np.float64) * 0.32
start -= start.min()
end = start + dur
print(f"blocks {st.size}  kernel span {end.max():.0f} us  mean block {dur.mean():.0f} us  p50 {np.percentile(dur,50):.0f}  p90 {np.percentile(dur,90):.0f}  p99 {np.percentile(dur,99):.0f}  max {dur.max():.0f}")
print(f"sum(block time)/span = average residency {dur.sum()/end.max():.0f} blocks ({dur.sum()/end.max()/256:.1f} per CU)")
# residency timeline
T = end.max(); bins = 24
edges = np.linspace(0, T, bins + 1)
res = [(np.minimum(end, edges[i+1]) - np.maximum(start, edges[i])).clip(0).sum() / (edges[i+1]-edges[i]) for i in range(bins)]
print("residency per time bin:", " ".join(f"{r:.0f}" for r in res))
# per position within the base archive (data classes are laid out in order)
per = dur.reshape(R, nb).mean(axis=0); csz = base["comp_size"].astype(np.float64)
seg = 16
for i in range(seg):
    a, b = i * nb // seg, (i + 1) * nb // seg
    print(f"  blocks {a:5d}-{b:5d}: mean {per[a:b].mean():6.0f} us  max {dur.reshape(R, nb)[:, a:b].max():6.0f} us  ratio {65536*(b-a)/csz[a:b].sum():5.2f}")
