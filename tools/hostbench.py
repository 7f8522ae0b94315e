#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer API (zxc_decompress / zxc_compress on pageable memory):
reported in DESIGN.md next to the HBM-resident number, never as the bench value."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import zxc_amd, oracle_py
from zxc_amd import corpus
data = corpus.synth_silesia(64 << 20, seed=0)
comp = oracle_py.Ref().compress(data, 3, 65536, True, False)
zxc_amd.decompress(comp)  # warm-up (context, scratch)
best = 1e9
for _ in range(5):
    t = time.perf_counter(); out = zxc_amd.decompress(comp); dt = time.perf_counter() - t; best = min(best, dt)
assert out == data
print(f"zxc_decompress host->host, {len(data)>>20} MiB: {len(data)/best/1e9:.2f} GB/s decoded ({best*1e3:.1f} ms)")
import ctypes as C
L = zxc_amd.lib(); dst = C.create_string_buffer(len(data))
best = 1e9
for _ in range(5):
    t = time.perf_counter(); rc = L.zxc_decompress(comp, len(comp), dst, len(data), None); dt = time.perf_counter() - t; best = min(best, dt)
assert rc == len(data)
print(f"  C call only (caller-owned buffers): {len(data)/best/1e9:.2f} GB/s decoded ({best*1e3:.1f} ms)")
best = 1e9
for _ in range(3):
    t = time.perf_counter(); c = zxc_amd.compress(data, 3, 65536, True); dt = time.perf_counter() - t; best = min(best, dt)
print(f"zxc_compress host->host level 3: {len(data)/best/1e9:.2f} GB/s source ({best*1e3:.1f} ms), ratio {len(data)/len(c):.3f}")
# FILE* caller: a 1 GiB archive from tmpfs, decoded into tmpfs and integrity-only (f_out = NULL); the read of batch i+1 overlaps
# the decode + write of batch i (zxc_stream_host.c)
if os.environ.get("HB_STREAM", "1") == "1":
    big = data * 16
    arc = "/dev/shm/zxc_hostbench.zxc"; outp = "/dev/shm/zxc_hostbench.out"
    with open(arc, "wb") as f: f.write(oracle_py.Ref().compress(big, 3, 65536, True, False))
    for label, dst in (("file -> file", outp), ("file -> integrity only", None)):
        zxc_amd.api.stream_decompress(arc, dst)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); rc = zxc_amd.api.stream_decompress(arc, dst); dt = time.perf_counter() - t; best = min(best, dt)
        assert rc == len(big), rc
        print(f"zxc_stream_decompress {label}, {len(big)>>20} MiB (tmpfs): {len(big)/best/1e9:.2f} GB/s decoded ({best*1e3:.0f} ms)")
    os.remove(arc)
    if os.path.exists(outp): os.remove(outp)
