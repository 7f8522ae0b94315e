#!/usr/bin/env python3
"""PCIe ceiling of the GPU box (what the host API's PCIe-inclusive rates are measured against): hipMemcpy of pinned and
pageable host buffers, each direction alone and both at once on two streams. GPU box only."""
import time
import torch

N = 1 << 30
dev = torch.device("cuda", 0)
d_a = torch.empty(N, dtype=torch.uint8, device=dev)
d_b = torch.empty(N, dtype=torch.uint8, device=dev)
h_pin_a = torch.empty(N, dtype=torch.uint8).pin_memory()
h_pin_b = torch.empty(N, dtype=torch.uint8).pin_memory()
h_page = torch.empty(N, dtype=torch.uint8)
h_page.fill_(1)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def best(fn, n=4):
    fn()
    torch.cuda.synchronize()
    b = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        b = min(b, time.perf_counter() - t0)
    return N / b / 1e9


print(f"H2D pinned   {best(lambda: d_a.copy_(h_pin_a, non_blocking=True)):7.1f} GB/s")
print(f"D2H pinned   {best(lambda: h_pin_b.copy_(d_b, non_blocking=True)):7.1f} GB/s")


def both():
    with torch.cuda.stream(s1):
        d_a.copy_(h_pin_a, non_blocking=True)
    with torch.cuda.stream(s2):
        h_pin_b.copy_(d_b, non_blocking=True)


print(f"H2D + D2H pinned at once: {best(both):7.1f} GB/s each direction")
print(f"H2D pageable {best(lambda: d_a.copy_(h_page)):7.1f} GB/s")
print(f"D2H pageable {best(lambda: h_page.copy_(d_b)):7.1f} GB/s")
