#!/usr/bin/env python3
"""bench.py — seekable decode throughput of the HIP block decoder on MI355X.

Metric (BASELINE.json): decode GB/s on silesia.tar-like data, level 3, seekable 64 KiB
independent blocks, compressed stream and block table already resident in HBM, output left
in HBM. One "step" = one launch of zxc_mi355x_decode_blocks_device over every block of the
workload on this rank's GPU. N GPUs: each rank owns its own contiguous block range of the
(virtually N-times larger) corpus, no collective on the data path -> weak scaling.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_workload(base_bytes, level, block_size, seed=0):
    """Synthetic silesia-like corpus compressed into a seekable level-`level` archive.

    The metric is defined on archives written by the reference encoder ("silesia.tar at -3"),
    so the untimed input preparation uses the unmodified reference compiled under oracle/_ref
    when it is there; otherwise it says so and stops (the device encoder is a later scope row).
    """
    from zxc_amd import corpus
    t0 = time.time()
    data = corpus.synth_silesia(base_bytes, seed=seed)
    t1 = time.time()
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    if not oracle_py.Ref.available():
        raise SystemExit("bench.py: oracle/_ref/libzxc_ref.so missing — cannot prepare a reference-encoded workload")
    ref = oracle_py.Ref()
    comp = ref.compress(data, level, block_size, True, False)
    t2 = time.time()
    return data, comp, dict(gen_s=round(t1 - t0, 2), compress_s=round(t2 - t1, 2), encoder="reference _ref")


def pmc_traffic(args, replicas):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/<round>_summary.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this same command). Only valid for
    the configuration the profile was taken on; otherwise null."""
    try:
        summ = json.load(open(os.path.join(ROOT, "profiles", "r1_summary.json")))
        if (args.base_mib, replicas, args.level, args.block_size) == (64, 32, 3, 65536):
            return summ["hbm_traffic_bytes_per_launch"]["total"]
    except Exception:
        pass
    return None


def cpu_baseline(comp, total, budget_s=12.0):
    """The reference's own parallel seekable decode on this box's host cores (bounded sample)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    cores = os.cpu_count() or 1
    if oracle_py.Ref.available():
        ref = oracle_py.Ref()
        dst = C.create_string_buffer(total)
        best = None
        t_start = time.time()
        iters = 0
        while iters < 2 or (time.time() - t_start < budget_s and iters < 40):
            t0 = time.perf_counter()
            rc, _ = ref.seekable_range_mt(comp, 0, total, cores, dst=dst)
            dt = time.perf_counter() - t0
            assert rc == total
            best = dt if best is None or dt < best else best
            iters += 1
        t0 = time.perf_counter()
        rc, _ = ref.seekable_range_mt(comp, 0, total, 1, dst=dst)
        dt1 = time.perf_counter() - t0
        return {"value": round(total / best / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "reference",
                "sample": f"zxc_seekable_decompress_range_mt over the {total >> 20} MiB base archive, "
                          f"T={cores} threads, best of {iters}",
                "single_thread_GBs": round(total / dt1 / 1e9, 3)}
    o = oracle_py.Oracle()
    n = min(total, 8 << 20)
    t0 = time.perf_counter()
    rc, _ = o.seekable_range(comp, 0, n)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"oracle C restatement, first {n >> 20} MiB, 1 thread"}


def main_encode(args):
    """configs[2]: per-block LZ77 match finding + GLO serialisation on the device, source resident in HBM,
    compressed blocks left in HBM. value = source GB/s; the output is round-trip checked (untimed) by
    decoding it on the device and comparing with the source."""
    import torch
    import zxc_amd
    from zxc_amd import corpus
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("ZXC_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("ZXC_BENCH_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    L = zxc_amd.lib()
    L.zxc_mi355x_set_device(local)
    dev = torch.device("cuda", local)
    bs = args.block_size
    data = corpus.synth_text(args.base_mib << 20, seed=1)
    tiles = max(1, args.replicas // 2)  # 64 MiB x 16 = 1 GiB of source per GPU by default
    d_src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev).repeat(tiles)
    n = d_src.numel()
    nb = (n + bs - 1) // bs
    stride = L.zxc_mi355x_encode_slot_stride(bs)
    d_slots = torch.empty(nb * stride, dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(nb, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        rc = L.zxc_mi355x_encode_blocks_device(C.c_void_p(d_src.data_ptr()), n, bs, args.level, 0,
                                               C.c_void_p(d_slots.data_ptr()), C.c_void_p(d_sizes.data_ptr()),
                                               C.c_void_p(stream))
        assert rc == 0, rc

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    # ---- round trip of what is being timed (first tile): host API compress -> device decode == source
    comp = zxc_amd.compress(data[:8 << 20], args.level, bs, True)
    assert zxc_amd.decompress(comp) == data[:8 << 20], "encoder output does not decode to the source"
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if rank == 0:
        csize = int(d_sizes.sum().item())
        kern_s = float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)])) / 1e3
        algo = n + csize
        print(json.dumps({
            "metric": "device LZ77 encode GB/s of source (enwik-like text, 64 KiB blocks, HBM-resident in/out)",
            "value": round(world * n * args.steps / wall / 1e9, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"configs[2]: synth_text {args.base_mib} MiB x {tiles} per GPU, level {args.level}, "
                                   f"{bs >> 10} KiB blocks, one wavefront per block", "blocks_per_gpu": nb,
                       "ratio": round(n / csize, 3), "parallelism": f"block-range x{world}, no collectives"},
            "roofline": {"bound": "hbm", "achieved": round(algo / kern_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(algo / kern_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "zxc_encode_blocks_kernel_h1x", "avg_launch_ms": round(kern_s * 1e3, 4),
                         "algorithmic_bytes_per_launch": algo},
            "round_trip": True}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--base-mib", type=int, default=int(os.environ.get("ZXC_BENCH_BASE_MIB", "64")))
    ap.add_argument("--replicas", type=int, default=int(os.environ.get("ZXC_BENCH_REPLICAS", "32")),
                    help="copies of the base archive resident in HBM at distinct addresses (defeats the 256 MiB LLC)")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--block-size", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=("decode", "encode"), default="decode",
                    help="decode = the headline metric (BASELINE.json configs[1]; --level 7 gives configs[4]); "
                         "encode = configs[2]: device match finder + serialiser over enwik-like text, GB/s of source")
    args = ap.parse_args()
    if args.mode == "encode":
        return main_encode(args)

    import torch
    import zxc_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ZXC_BENCH_BACKEND=gloo + ZXC_BENCH_DEVICE=0 lets a 1-GPU box rehearse the N-rank path (all ranks on one GPU)
    backend = os.environ.get("ZXC_BENCH_BACKEND", "nccl")
    if "ZXC_BENCH_DEVICE" in os.environ:
        local = int(os.environ["ZXC_BENCH_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    zxc_amd.lib().zxc_mi355x_set_device(local)
    dev = torch.device("cuda", local)

    # ---- workload (same bytes on every rank; each rank decodes its own replicas = its block range)
    data, comp, prep = build_workload(args.base_mib << 20, args.level, args.block_size)
    s = zxc_amd.Seekable(comp)
    nb = s.num_blocks
    base_jobs = s.plan()
    total = s.decompressed_size
    R = args.replicas
    comp_stride = (len(comp) + 255) & ~255
    out_stride = (total + 255) & ~255
    d_comp = torch.empty(R * comp_stride + 256, dtype=torch.uint8, device=dev)
    h_comp = torch.frombuffer(bytearray(comp), dtype=torch.uint8)
    for r in range(R):
        d_comp[r * comp_stride: r * comp_stride + len(comp)].copy_(h_comp)
    jobs = np.tile(base_jobs, R)
    rep = np.repeat(np.arange(R, dtype=np.uint64), nb)
    jobs["comp_off"] += rep * np.uint64(comp_stride)
    jobs["out_off"] += rep * np.uint64(out_stride)
    n_jobs = int(jobs.size)
    d_jobs = torch.frombuffer(bytearray(jobs.tobytes()), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(R * out_stride + 256, dtype=torch.uint8, device=dev)
    d_status = torch.full((n_jobs,), -999, dtype=torch.int32, device=dev)
    algo_bytes = int(jobs["comp_size"].astype(np.int64).sum() + jobs["out_len"].astype(np.int64).sum())
    out_bytes = int(jobs["out_len"].astype(np.int64).sum())
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        zxc_amd.decode_blocks_device(d_comp.data_ptr(), d_jobs.data_ptr(), n_jobs, d_out.data_ptr(),
                                     d_status.data_ptr(), args.block_size, False, stream)

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    # ---- bit-exactness of what is being timed: every block status, and every replica's bytes
    st = d_status.cpu().numpy()
    assert (st == jobs["out_len"].astype(np.int32)).all(), f"block status mismatch: {st[st != jobs['out_len']][:8]}"
    want = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    for r in range(R):
        assert torch.equal(d_out[r * out_stride: r * out_stride + total], want), f"replica {r} differs from the input corpus"
    del want

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    kern_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    if rank == 0:
        avg_kernel_s = float(np.mean(kern_ms)) / 1e3
        value = world * out_bytes * args.steps / wall / 1e9
        achieved = algo_bytes / avg_kernel_s / 1e9
        line = {
            "metric": "seekable decode GB/s (level 3, 64 KiB independent blocks, HBM-resident in/out)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"configs[1]: synth_silesia {args.base_mib} MiB x {R} HBM replicas per GPU, "
                                   f"level {args.level}, {args.block_size >> 10} KiB seekable blocks, one wavefront per block",
                       "blocks_per_gpu": n_jobs, "decoded_bytes_per_gpu": out_bytes, "compressed_bytes_per_gpu": algo_bytes - out_bytes,
                       "ratio": round(out_bytes / (algo_bytes - out_bytes), 3), "parallelism": f"block-range x{world}, no collectives",
                       "prep": prep},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(args, R),
                         "kernel": "zxc_decode_blocks_kernel", "avg_launch_ms": round(avg_kernel_s * 1e3, 4),
                         "algorithmic_bytes_per_launch": algo_bytes},
            "bit_exact": True,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(comp, total)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
