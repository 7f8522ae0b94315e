#!/usr/bin/env python3
"""Measured HBM copy rate on this GPU (read + write bytes per second of a large device-to-device copy):
context for the roofline fraction, which is quoted against the 8 TB/s spec peak."""
import torch
dev = torch.device("cuda", 0)
n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
a.fill_(7); b.copy_(a); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"d2d copy of {n>>30} GiB: {ms:.3f} ms -> {2*n/ms/1e6:.0f} GB/s read+write ({n/ms/1e6:.0f} GB/s each way)")
e0.record()
for _ in range(10): a.fill_(3)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"fill of {n>>30} GiB: {ms:.3f} ms -> {n/ms/1e6:.0f} GB/s write")
