#!/usr/bin/env python3
"""A/B of library variants on the bench workload (GPU box only): python tools/abbench.py libA.so libB.so ...
Builds the workload once per process (bench.build_workload), times the decode launch, checks the bytes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, zxc_amd, bench
    mib = int(os.environ.get("AB_MIB", "64")); R = int(os.environ.get("AB_REPL", "32")); level = int(os.environ.get("AB_LEVEL", "3"))
    data, comp, prep = bench.build_workload(mib << 20, level, 65536)
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
    ok = bool((d_st.cpu().numpy() == jobs["out_len"].astype(np.int32)).all()) and bytes(d_out[(R-1)*osz:(R-1)*osz+total].cpu().numpy()) == data
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): step()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print(f"{os.environ.get('ZXC_LIB_VARIANT','libzxc_mi355x.so'):24s} L{level} ok={ok} {best:7.3f} ms {R*total/best/1e6:8.1f} GB/s", flush=True)
else:
    for lib in sys.argv[1:]:
        env = dict(os.environ); env["ZXC_LIB_VARIANT"] = lib
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env)
