#!/usr/bin/env python3
"""bench.py — seekable decode throughput of the HIP block decoder on MI355X.

Metric (BASELINE.json): decode GB/s on silesia.tar-like data, level 3, seekable 64 KiB
independent blocks, compressed stream and block table already resident in HBM, output left
in HBM. The workload is ONE seekable corpus of N x --tiles unique silesia-mix tiles behind ONE seek
table; rank g of N decodes the contiguous block range [g*B//N, (g+1)*B//N) it gets from
zxc_mi355x_plan_seekable(first, n, comp_rebase) — no collective on the data path, per-GPU work fixed
as N grows -> weak scaling. One "step" = one launch of zxc_mi355x_decode_blocks_device over the
rank's whole range.

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


def archive_parts(comp, block_size):
    """A seekable archive written by the reference -> (its blocks region, comp_sizes[] of its seek table,
    its 16-byte file header, its 8-byte EOF block). Layout: docs/FORMAT.md §3-§9 (header 16 B, blocks, EOF
    block, SEK block = 8-byte header + 4 B per block, 12-byte footer)."""
    total = int.from_bytes(comp[-12:-4], "little")
    nb = (total + block_size - 1) // block_size
    sizes = np.frombuffer(comp, dtype="<u4", count=nb, offset=len(comp) - 12 - 4 * nb).astype(np.uint32)
    end = 16 + int(sizes.astype(np.int64).sum())
    return memoryview(comp)[16:end], sizes, bytes(comp[:16]), bytes(comp[end:end + 8])


def tile_bytes(tile, block_size, pool):
    """Plaintext of corpus tile `tile` (a whole number of blocks): the synthetic silesia mix with the tile's own
    xor-rotated seed, or the real silesia.tar (ZXC_CORPUS_DIR) rotated by a per-tile byte shift."""
    from zxc_amd import corpus
    n = (corpus.TILE_BYTES // block_size) * block_size
    d = os.environ.get("ZXC_CORPUS_DIR")
    if d and os.path.exists(os.path.join(d, "silesia.tar")):
        raw = open(os.path.join(d, "silesia.tar"), "rb").read()
        sh = (tile * 104729 * 64) % len(raw)
        return (raw[sh:] + raw[:sh])[:n], "silesia.tar (ZXC_CORPUS_DIR), rotated per tile"
    return corpus.synth_silesia_tile(tile, pool=pool)[:n], "synth_silesia tiles (per-tile xor-rotated seed)"


CPU_BASELINE_TILES = 4  # the cpu_baseline sample: ONE reference-written archive of this many corpus tiles


def _ref_compress_cached(ref, key, data, level, block_size, seekable, checksum):
    """ref.compress, kept on disk when ZXC_BENCH_CACHE names a directory (tools/profile.sh: its five rocprofv3 passes run the same
    command on the same box; the reference encoder is untimed input preparation, 100 s per pass at level 7). Off by default."""
    d = os.environ.get("ZXC_BENCH_CACHE")
    if not d:
        return ref.compress(data, level, block_size, seekable, checksum)
    os.makedirs(d, exist_ok=True)
    import hashlib
    path = os.path.join(d, f"{hashlib.sha1(key.encode()).hexdigest()[:16]}_n{len(data)}_l{level}_b{block_size}_s{int(seekable)}_c{int(checksum)}.zxc")
    if os.path.exists(path):
        return open(path, "rb").read()
    comp = ref.compress(data, level, block_size, seekable, checksum)
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "wb") as f:
        f.write(comp)
    os.replace(tmp, path)
    return comp


def build_rank_corpus(first, last, level, block_size, pool, dev, checksum=False, baseline_tiles=0):
    """Blocks [first, last) of the global corpus, encoded by the unmodified reference (the metric is defined on
    archives written by the reference encoder: "silesia.tar at -3"; untimed input preparation). Only the tiles
    this range touches are generated; compressed blocks and plaintext go straight to HBM, tile by tile, so the
    host never holds more than a few tiles. -> (d_comp, comp_sizes of the range, d_want, header16, eof8, info)"""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from zxc_amd import corpus
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    if not oracle_py.Ref.available():
        raise SystemExit("bench.py: oracle/_ref/libzxc_ref.so missing — cannot prepare a reference-encoded workload")
    ref = oracle_py.Ref()
    tb = corpus.TILE_BYTES // block_size           # blocks per tile
    tiles = list(range(first // tb, (last - 1) // tb + 1)) if last > first else []
    t0 = time.time()
    regions, sizes_all, wants = [], [], []
    hdr = eof = None
    first_tile_comp = None
    src = ""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    base_fut, base_data = None, []
    with ThreadPoolExecutor(max_workers=max(1, min(16, len(tiles) + 1, (os.cpu_count() or 1) // world // 2))) as tp:
        futs = []
        for t in tiles:
            data, src = tile_bytes(t, block_size, pool)
            if len(base_data) < baseline_tiles:  # the CPU baseline's archive: the first tiles as ONE seekable archive
                base_data.append(data)
                if len(base_data) == min(baseline_tiles, len(tiles)):
                    base_fut = tp.submit(_ref_compress_cached, ref, f"base{tiles[0]}x{len(base_data)}_{src}", b"".join(base_data), level, block_size, True, False)
            lo, hi = max(first, t * tb) - t * tb, min(last, (t + 1) * tb) - t * tb
            wants.append(torch.frombuffer(bytearray(data[lo * block_size: hi * block_size]), dtype=torch.uint8).to(dev))
            futs.append((lo, hi, tp.submit(_ref_compress_cached, ref, f"tile{t}_{src}", data, level, block_size, True, checksum)))
            del data
        for lo, hi, f in futs:
            comp = f.result()
            if first_tile_comp is None:
                first_tile_comp = comp
            region, sizes, hdr, eof = archive_parts(comp, block_size)
            off = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
            regions.append(torch.frombuffer(bytearray(region[off[lo]: off[hi]]), dtype=torch.uint8).to(dev))
            sizes_all.append(sizes[lo:hi])
    d_comp = torch.cat(regions + [torch.zeros(256, dtype=torch.uint8, device=dev)])  # (+ slack for 16-byte reads)
    d_want = torch.cat(wants)
    info = dict(prep_s=round(time.time() - t0, 1), tiles=len(tiles), encoder="reference _ref", source=src)
    if base_fut is not None:
        first_tile_comp = base_fut.result()
        info["prep_s"] = round(time.time() - t0, 1)
    return d_comp, np.concatenate(sizes_all), d_want, hdr, eof, info, first_tile_comp


def open_global_table(all_sizes, block_size, hdr, eof, total_decoded):
    """ONE seek table for the whole corpus: the archive is opened through zxc_seekable_open_reader over a
    reader that serves the file header, the EOF block, the SEK block (built by zxc_write_seek_table from the
    comp_sizes of ALL ranks) and the footer — the block bytes themselves stay in each rank's HBM."""
    import zxc_amd
    nb = int(all_sizes.size)
    L = zxc_amd.lib()
    L.zxc_seek_table_size.restype = C.c_size_t
    L.zxc_write_seek_table.restype = C.c_int64
    L.zxc_write_seek_table.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32]
    sek = C.create_string_buffer(int(L.zxc_seek_table_size(nb)))
    cs = np.ascontiguousarray(all_sizes, dtype=np.uint32)
    assert L.zxc_write_seek_table(sek, len(sek), cs.ctypes.data, nb) == len(sek)
    blocks_bytes = int(cs.astype(np.int64).sum())
    tail = eof + sek.raw + int(total_decoded).to_bytes(8, "little") + (0).to_bytes(4, "little")
    tail_off = 16 + blocks_bytes
    size = tail_off + len(tail)

    def reader(off, n):
        if off + n <= 16:
            return hdr[off:off + n]
        if off >= tail_off and off + n <= size:
            return tail[off - tail_off: off - tail_off + n]
        return None  # block bytes are device-resident; nothing at open time asks for them
    return zxc_amd.Seekable(reader=reader, size=size)


def cpu_baseline(comp, total, budget_s=12.0, what="corpus tile 0"):
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
                "sample": f"zxc_seekable_decompress_range_mt over {what} ({total >> 20} MiB decoded, "
                          f"{(total + 65535) >> 16} blocks of 64 KiB), T={cores} threads, best of {iters}",
                "single_thread_GBs": round(total / dt1 / 1e9, 3)}
    o = oracle_py.Oracle()
    n = min(total, 8 << 20)
    t0 = time.perf_counter()
    rc, _ = o.seekable_range(comp, 0, n)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"oracle C restatement, first {n >> 20} MiB, 1 thread"}


def encode_run(args, level, enc_mib, steps, warmup, comm, with_cpu_baseline=True, block_size=None, extras=True):
    """configs[2]: per-block LZ77 hash-chain match finding + GLO serialisation on the device over enc_mib MiB of unique
    enwik-like text per GPU, source resident in HBM, compressed blocks left in HBM. value = source GB/s. The
    compressed stream of the LAST timed launch is round-tripped: every block decoded on the device and compared
    with the source (all of it), and a 64 MiB sample decoded by the unmodified reference decoder on the host.
    -> the result line (a dict) on rank 0, None elsewhere."""
    import multiprocessing as mp
    import torch
    import zxc_amd
    from zxc_amd import corpus
    rank, world, local, backend, dist = comm
    pool = mp.get_context("spawn").Pool(max(1, min(32, (os.cpu_count() or 1) // world)))
    L = zxc_amd.lib()
    L.zxc_mi355x_gather_blocks_device.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    dev = torch.device("cuda", local)
    bs = block_size or args.block_size
    n = enc_mib << 20
    t0 = time.time()
    parts = pool.map(corpus.gen_chunk, corpus.enwik_chunks(n, seed=1 + rank))
    pool.close()
    d_src = torch.cat([torch.frombuffer(bytearray(p), dtype=torch.uint8).to(dev) for p in parts] +
                      [torch.zeros(256, dtype=torch.uint8, device=dev)])
    sample = b"".join(parts[:8])  # 64 MiB kept on the host for the reference-decoder leg and the CPU baseline
    del parts
    prep_s = round(time.time() - t0, 1)
    nb = (n + bs - 1) // bs
    stride = L.zxc_mi355x_encode_slot_stride(bs)
    d_slots = torch.empty(nb * stride, dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(nb, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        rc = L.zxc_mi355x_encode_blocks_device(C.c_void_p(d_src.data_ptr()), n, bs, level, 0,
                                               C.c_void_p(d_slots.data_ptr()), C.c_void_p(d_sizes.data_ptr()),
                                               C.c_void_p(stream))
        assert rc == 0, rc

    for _ in range(max(warmup, 1)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    # ---- round trip of what was timed: compact the slots (seek-table entries = d_sizes), decode every block on
    # the device, compare with the source
    sizes = d_sizes.to(torch.int64)
    offs = torch.cumsum(sizes, 0) - sizes
    csize = int(sizes.sum().item())
    d_offs = offs.contiguous()
    d_comp = torch.zeros(csize + 256, dtype=torch.uint8, device=dev)
    rc = L.zxc_mi355x_gather_blocks_device(C.c_void_p(d_slots.data_ptr()), bs, C.c_void_p(d_sizes.data_ptr()),
                                           C.c_void_p(d_offs.data_ptr()), C.c_void_p(d_comp.data_ptr()), nb, C.c_void_p(stream))
    assert rc == 0
    jobs = np.zeros(nb, dtype=zxc_amd.api.JOB_DTYPE)
    jobs["comp_off"] = offs.cpu().numpy().astype(np.uint64)
    jobs["comp_size"] = sizes.cpu().numpy().astype(np.uint32)
    jobs["out_off"] = np.arange(nb, dtype=np.uint64) * bs
    jobs["out_len"] = np.minimum(bs, n - np.arange(nb, dtype=np.int64) * bs).astype(np.uint32)
    d_jobs = torch.frombuffer(bytearray(jobs.tobytes()), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(n + 256, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(nb, dtype=torch.int32, device=dev)
    zxc_amd.decode_blocks_device(d_comp.data_ptr(), d_jobs.data_ptr(), nb, d_out.data_ptr(), d_st.data_ptr(), bs, False, stream)
    torch.cuda.synchronize()
    assert torch.equal(d_st, torch.from_numpy(jobs["out_len"].astype(np.int32)).to(dev)), "a block written by the encoder does not decode"
    assert torch.equal(d_out[:n], d_src[:n]), "encoder output does not decode to the source"
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    ref_ok = None
    if oracle_py.Ref.available():  # the unmodified reference decoder accepts the stream (host API: same kernels + framing)
        comp = zxc_amd.compress(sample, level, bs, True)
        rc, out = oracle_py.Ref().decompress(comp, len(sample))
        assert rc == len(sample) and out == sample, "reference decoder rejects the encoder's archive"
        ref_ok = f"{len(sample) >> 20} MiB archive decoded bit-exact by the unmodified reference decoder"
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if rank != 0:
        return None
    kern_s = float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])) / 1e3
    algo = n + csize
    entry = {1: "l1", 2: "l2", 3: "l3", 4: "l4", 5: "l57"}.get(level, "l67")
    if bs > 65536 and 3 <= level <= 5:
        entry = "l67"  # (zxc_enc_level_bs, zxc_encode_levels.h: blocks above 64 KiB take the entry with the 2^15-position chain ring)
    line = {
        "metric": f"device LZ77 hash-chain encode GB/s of source (level {level}, enwik-like text, {bs >> 10} KiB blocks, HBM-resident in/out)",
        "value": round(world * n * steps / wall / 1e9, 2), "unit": "GB/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(wall / steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"configs[2]: {enc_mib} MiB of unique enwik-like text per GPU (synth chunks of 8 MiB, own seed "
                               f"each), level {level}, {bs >> 10} KiB blocks, one wavefront per block", "blocks_per_gpu": nb,
                   "ratio": round(n / csize, 3), "parallelism": f"block-range x{world}, no collectives", "prep_s": prep_s},
        "roofline": {"bound": "hbm", "achieved": round(algo / kern_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(algo / kern_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": profiled_traffic("encode_l%d" % level, enc_mib=enc_mib, block_size=bs),
                     "kernel": f"zxc_encode_blocks_kernel_{entry}", "avg_launch_ms": round(kern_s * 1e3, 4),
                     "algorithmic_bytes_per_launch": algo},
        "round_trip": {"device": f"all {nb} blocks decoded on the device == source", "reference": ref_ok}}
    if not extras:
        return line
    # incompressible input (where the reference's match finder accelerates its steps, src/lib/zxc_compress.c:1176): 256 MiB of random bytes,
    # every block must come out RAW (stored: 8-byte header + the bytes) and decode back
    try:
        rn = 256 << 20
        d_rnd = torch.randint(0, 256, (rn + 256,), dtype=torch.uint8, device=dev)
        rnb = rn // bs
        def rstep():
            assert L.zxc_mi355x_encode_blocks_device(C.c_void_p(d_rnd.data_ptr()), rn, bs, level, 0, C.c_void_p(d_slots.data_ptr()),
                                                     C.c_void_p(d_sizes.data_ptr()), C.c_void_p(stream)) == 0
        rstep(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); rstep(); rstep(); e1.record(); torch.cuda.synchronize()
        rms = e0.elapsed_time(e1) / 2
        raw_ok = bool((d_sizes[:rnb] == bs + 8).all().item())
        slots = d_slots[:rnb * stride].view(rnb, stride)
        raw_ok = raw_ok and torch.equal(slots[:, 8:8 + bs].reshape(-1), d_rnd[:rn])
        line["incompressible"] = {"value": round(rn / rms / 1e6, 1), "unit": "GB/s of source", "bytes": rn, "ms": round(rms, 3),
                                  "all_blocks_raw_and_equal_to_the_source": raw_ok,
                                  "what": "256 MiB of random bytes: after 8 chunks (512 B) without a sequence only every fourth chunk of 64 positions is searched"}
        assert raw_ok, "incompressible blocks must be stored RAW"
        del d_rnd
    except AssertionError:
        raise
    except Exception as ex:  # (never lose the line over the extra measurement)
        line["incompressible"] = {"error": repr(ex)}
    if with_cpu_baseline and world == 1 and oracle_py.Ref.available():
        line["cpu_baseline"] = cpu_baseline_encode(sample, level, bs)
    return line


def cpu_baseline_encode(sample, level, bs):
    """The reference's zxc_compress on this box's host cores: one call per thread over equal slices of a bounded
    sample (zxc_compress itself is single-threaded; ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle_py
    ref = oracle_py.Ref()
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    c1 = ref.compress(sample[:16 << 20], level, bs, True, False)
    dt1 = time.perf_counter() - t0
    T = min(cores, 64)
    sl = len(sample) // T
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=T) as tp:
        outs = list(tp.map(lambda k: len(ref.compress(sample[k * sl:(k + 1) * sl], level, bs, True, False)), range(T)))
    dt = time.perf_counter() - t0
    return {"value": round(T * sl / dt / 1e9, 3), "unit": "GB/s", "cores": T, "kind": "reference",
            "sample": f"zxc_compress level {level} over {T} slices of {sl >> 10} KiB of the same text, one thread each",
            "single_thread_GBs": round((16 << 20) / dt1 / 1e9, 3), "ratio": round(T * sl / sum(outs), 3),
            "ratio_1t_16MiB": round((16 << 20) / len(c1), 3)}


def calibration_launch(dev, bs, n=16384):
    """One launch of the decode kernel over n RAW blocks (8-byte header + bs stored bytes each): the kernel's
    RAW path is a pure 16 B/lane copy, so this launch reads n*(bs+8) and writes n*bs bytes — known numbers in
    the same rocprofv3 pass as the real launches (MI355X_MICROARCH.md: FETCH_SIZE needs calibrating)."""
    import torch
    import zxc_amd
    hdr = bytearray(8)  # type 0 = RAW, comp_size = bs; the header check byte is the frame walker's business, not the kernel's
    hdr[3:7] = bs.to_bytes(4, "little")
    t = torch.randint(0, 256, (n, bs + 8), dtype=torch.uint8, device=dev)
    t[:, :8] = torch.tensor(list(hdr), dtype=torch.uint8, device=dev)
    jobs = np.zeros(n, dtype=zxc_amd.api.JOB_DTYPE)
    jobs["comp_off"] = np.arange(n, dtype=np.uint64) * (bs + 8)
    jobs["out_off"] = np.arange(n, dtype=np.uint64) * bs
    jobs["comp_size"] = bs + 8
    jobs["out_len"] = bs
    d_jobs = torch.frombuffer(bytearray(jobs.tobytes()), dtype=torch.uint8).to(dev)
    d_o = torch.zeros(n * bs + 256, dtype=torch.uint8, device=dev)
    d_s = torch.zeros(n, dtype=torch.int32, device=dev)
    zxc_amd.decode_blocks_device(t.data_ptr(), d_jobs.data_ptr(), n, d_o.data_ptr(), d_s.data_ptr(), bs, False,
                                 torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bool((d_s == bs).all().item()) and torch.equal(d_o[:n * bs].view(n, bs), t[:, 8:])
    return {"dispatch": "first zxc_decode_blocks_kernel launch", "read_bytes": n * (bs + 8), "write_bytes": n * bs}


def init_ranks(cpu=False):
    """(rank, world, local device, backend, dist or None). One process per GPU; the driver launches N > 1 through
    torch.distributed.run. ZXC_BENCH_BACKEND=gloo + ZXC_BENCH_DEVICE=0 rehearses the N-rank path on one GPU."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("ZXC_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("ZXC_BENCH_BACKEND", "nccl")
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl" and not cpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local, backend, dist


def corpus_tiles(world, tiles_per_gpu, total_tiles=0):
    """Tiles of the ONE corpus: weak scaling = world x tiles_per_gpu (per-GPU work fixed); strong scaling (--scaling strong,
    BASELINE configs[3] as worded: ONE 64 GiB corpus at 1/2/4/8 GPUs) = total_tiles whatever the world size."""
    return total_tiles if total_tiles else world * tiles_per_gpu


def rank_partition(rank, world, tiles_per_gpu, block_size, total_tiles=0):
    """The north-star partition: ONE corpus of N blocks in one seek table; rank g decodes the contiguous index range
    [g*N//G, (g+1)*N//G) (zxc_amd.shard.block_range)."""
    from zxc_amd import corpus, shard
    n_total = corpus_tiles(world, tiles_per_gpu, total_tiles) * (corpus.TILE_BYTES // block_size)
    first, last = shard.block_range(rank, world, n_total)
    return n_total, first, last


PREP_CORE_S_PER_TILE = {1: 3.0, 2: 3.0, 3: 5.0, 4: 8.0, 5: 25.0, 6: 120.0, 7: 160.0}  # generation + reference encode of one 202 MiB tile, one core


def prep_estimate_s(n_tiles, level, world):
    """Rough wall time of this rank's input preparation (synthetic tiles + the reference encoder, untimed): every rank works on
    1/world of the box's cores, at most 16 tiles at once (build_rank_corpus)."""
    workers = max(1, min(16, n_tiles + 1, (os.cpu_count() or 1) // world // 2))
    return -(-n_tiles // workers) * PREP_CORE_S_PER_TILE.get(level, 5.0)


def decode_run(args, level, tiles, steps, warmup, comm, checksum=False, calib_launch=False, with_cpu_baseline=True,
               cpu_budget_s=12.0, block_size=None):
    """One decode measurement: this rank's block range of ONE corpus of world x tiles tiles, `steps` timed launches
    bracketed by barrier + synchronize, every byte and status checked before and after. -> the result line on rank 0."""
    import multiprocessing as mp
    import torch
    import zxc_amd
    rank, world, local, backend, dist = comm
    # generator processes (numpy only; spawn keeps them free of torch state)
    pool = mp.get_context("spawn").Pool(max(1, min(48, (os.cpu_count() or 1) // world)))
    dev = torch.device("cuda", local)
    bs = block_size or args.block_size
    strong = getattr(args, "scaling", "weak") == "strong"
    total_tiles = (args.total_tiles or 8 * tiles) if strong else 0
    all_tiles = corpus_tiles(world, tiles, total_tiles)

    # ---- workload: this rank's block range of the one corpus
    n_total, first, last = rank_partition(rank, world, tiles, bs, total_tiles)
    from zxc_amd import corpus as _corpus
    my_tiles = (last - 1) // (_corpus.TILE_BYTES // bs) - first // (_corpus.TILE_BYTES // bs) + 1 if last > first else 0
    est = prep_estimate_s(my_tiles, level, world)
    print(f"[bench] rank {rank}/{world}: blocks [{first}, {last}) of {n_total} ({my_tiles} tiles, level {level}, {bs >> 10} KiB blocks); "
          f"input preparation (reference encoder, untimed) estimated at ~{est:.0f} s", file=sys.stderr, flush=True)
    if est > args.max_prep_s:
        raise SystemExit(f"bench.py: preparing {my_tiles} tiles per rank at level {level} would take ~{est:.0f} s (> --max-prep-s "
                         f"{args.max_prep_s}): lower --tiles / --total-tiles or raise --max-prep-s")
    base_tiles = min(CPU_BASELINE_TILES, tiles) if (with_cpu_baseline and world == 1) else 0
    d_comp, my_sizes, d_want, hdr, eof, prep, tile0_comp = build_rank_corpus(first, last, level, bs, pool, dev, checksum, base_tiles)
    pool.close()
    if world > 1:  # control plane only: every rank learns the whole seek table (4 B per block)
        gathered = [None] * world
        dist.all_gather_object(gathered, my_sizes.tobytes())
        all_sizes = np.concatenate([np.frombuffer(b, dtype=np.uint32) for b in gathered])
    else:
        all_sizes = my_sizes
    assert all_sizes.size == n_total
    s = open_global_table(all_sizes, bs, hdr, eof, n_total * bs)
    assert s.num_blocks == n_total and s.decompressed_size == n_total * bs
    comp_rebase = 16 + int(all_sizes[:first].astype(np.int64).sum())
    jobs = s.plan(first, last - first, comp_rebase)  # zxc_mi355x_plan_seekable(first, n, comp_rebase)
    n_jobs = int(jobs.size)
    d_jobs = torch.frombuffer(bytearray(jobs.tobytes()), dtype=torch.uint8).to(dev)
    out_bytes = int(jobs["out_len"].astype(np.int64).sum())
    algo_bytes = int(jobs["comp_size"].astype(np.int64).sum()) + out_bytes
    d_out = torch.zeros(out_bytes + 256, dtype=torch.uint8, device=dev)
    d_status = torch.full((n_jobs,), -999, dtype=torch.int32, device=dev)
    want_status = torch.from_numpy(jobs["out_len"].astype(np.int32)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        zxc_amd.decode_blocks_device(d_comp.data_ptr(), d_jobs.data_ptr(), n_jobs, d_out.data_ptr(),
                                     d_status.data_ptr(), bs, checksum, stream)

    def check(when):
        assert torch.equal(d_status, want_status), f"{when}: block status mismatch on rank {rank}"
        assert torch.equal(d_out[:out_bytes], d_want), f"{when}: decoded bytes differ from the corpus on rank {rank}"

    calib = None
    if calib_launch:
        calib = calibration_launch(dev, bs)
    for _ in range(max(warmup, 1)):
        step()
    torch.cuda.synchronize()
    check("before timing")  # bit-exactness of what is being timed: every block's status and every byte

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    kern_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    d_out.zero_()
    d_status.fill_(-999)
    step()
    torch.cuda.synchronize()
    check("after timing")
    # every rank's own numbers (an imbalance must be visible): control plane only, after the timed region
    # (world_seen / backend / device: what THIS rank's process group and device really were — the driver's N-GPU line can be checked
    #  for "N ranks, N distinct devices" from the line alone)
    mine = {"rank": rank, "blocks": [int(first), int(last)], "decoded_bytes": out_bytes, "wall_s": round(wall, 6),
            "GBs": round(out_bytes * steps / wall / 1e9, 2), "avg_launch_ms": round(float(np.mean(kern_ms)), 4), "prep_s": prep["prep_s"],
            "world_seen": int(dist.get_world_size()) if (dist is not None and world > 1) else 1, "backend": backend,
            "device": int(torch.cuda.current_device()), "device_name": torch.cuda.get_device_name(torch.cuda.current_device()),
            "pci_bus": getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "pci_bus_id", None)}
    if world > 1:
        t = torch.tensor([wall, float(out_bytes)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        wall = float(tmax[0].item())
        total_out = int(t[1].item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    else:
        total_out = out_bytes
        per_rank = [mine]
    if rank != 0:
        return None
    avg_kernel_s = float(np.mean(kern_ms)) / 1e3
    value = total_out * steps / wall / 1e9
    achieved = algo_bytes / avg_kernel_s / 1e9
    cfg = "configs[4]" if level == 7 else "configs[1]" if level == 3 else f"configs[1] at level {level}"
    if bs != 65536:
        cfg += f" at the reference's {'default' if bs == 524288 else 'maximum' if bs == 2097152 else 'other'} block size"
    if world > 1 or strong:
        cfg = f"configs[3] ({all_tiles * 211943424 / 2**30:.1f} GiB corpus over {world} GPUs, {'strong' if strong else 'weak'} scaling)"
    traffic = None
    if not (checksum or strong or world > 1):
        if getattr(args, "live_traffic", False) and level == 3:
            del d_comp, d_want, d_out  # (the child needs a few GB of its own; this run's device buffers are done)
            torch.cuda.empty_cache()
            d_comp = d_want = d_out = None
            traffic = measure_traffic_live(level, bs, n_jobs, algo_bytes)
        if traffic is None:
            traffic = profiled_traffic("decode_l%d" % level, tiles=tiles, block_size=bs)
    line = {
        "metric": f"seekable decode GB/s (level {level}, {bs >> 10} KiB independent blocks, HBM-resident in/out"
                  + (", per-block checksums verified on the device)" if checksum else ")"),
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(wall / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{cfg}: ONE seekable corpus of {all_tiles} tiles x 211943424 B ({prep['source']}), "
                               f"level {level}, {bs >> 10} KiB blocks, {n_total} blocks in one seek table; rank g decodes "
                               f"blocks [g*N//G, (g+1)*N//G) (rank 0: [{first}, {last})), one wavefront per block",
                   "blocks_per_gpu": n_jobs, "decoded_bytes_per_gpu": out_bytes,
                   "compressed_bytes_per_gpu": algo_bytes - out_bytes,
                   "ratio": round(out_bytes / (algo_bytes - out_bytes), 3),
                   "parallelism": f"seek-table block range x{world}, no collectives on the data path", "prep": prep,
                   "per_rank": per_rank},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "kernel": "zxc_decode_blocks_lean_kernel + zxc_decode_blocks_kernel (blocks with coded sections), side by side"
                               if level <= 5 else "zxc_pivco_sections_{small,medium,large}_kernel (coded sections -> scratch), then "
                                                  "zxc_decode_blocks_lean_pre_kernel; zxc_decode_blocks_kernel beside them for the rest",
                     "avg_launch_ms": round(avg_kernel_s * 1e3, 4), "algorithmic_bytes_per_launch": algo_bytes},
        "bit_exact": "every byte and block status checked before and after the timed loop",
    }
    if calib:
        line["calibration"] = calib
    if with_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(tile0_comp, int.from_bytes(tile0_comp[-12:-4], "little"), cpu_budget_s,
                                            f"corpus tiles 0-{base_tiles - 1} as one archive" if base_tiles > 1 else "corpus tile 0")
    d_comp = d_want = d_out = None
    torch.cuda.empty_cache()
    return line


# The device sources a traffic profile belongs to, by explicit list (renaming or adding a HOST file under zxc_amd/csrc
# must not change the hash, and a profile of the decoder stays valid when only the encoder changes). A/B-only sources
# (zxc_seq_own.inc: compiled under -DLEAN_OWNER alone) are not part of the product's kernels.
DEVICE_SOURCES = {
    "decode": ("zxc_decode_kernel.hip", "zxc_seq_lean.inc", "zxc_pivco.inc", "zxc_pivco_dir.inc", "zxc_rapidhash.inc", "zxc_dev.h", "zxc_lds.h",
               "zxc_experiments.h", "zxc_hip_shim.hip"),
    "encode": ("zxc_encode_kernel.hip", "zxc_optparse.inc", "zxc_pivco_encode.inc", "zxc_rapidhash.inc", "zxc_encode_levels.h", "zxc_dev.h",
               "zxc_experiments.h", "zxc_hip_shim.hip"),
}


def kernel_sources_hash(what="decode"):
    """sha256 over the device sources of the decoder / the encoder (DEVICE_SOURCES, in that order): what a traffic profile belongs to.
    (Content hash, not a git object id: .git does not travel to the GPU box.)"""
    import hashlib
    d = os.path.join(ROOT, "zxc_amd", "csrc")
    h = hashlib.sha256()
    for f in DEVICE_SOURCES["encode" if what.startswith("encode") else "decode"]:
        h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read() + b"\0")
    return h.hexdigest()[:16]


def profiled_traffic(what, **workload):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload (FETCH_SIZE / WRITE_SIZE in
    separate --pmc runs, counters corrected as profiles/r3_gather_calibration.log prescribes; tools/profile.sh +
    tools/profile_summary.py write profiles/r6_traffic.json — nobody else does). NOT measured in this run: a constant, reported only when the
    run's workload equals the profiled one AND the device sources are byte for byte the ones that were profiled
    (`kernels` = kernel_sources_hash() at profiling time) — else null, never a stale number."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "r6_traffic.json")))
        e = tab.get(what + "_bs%d" % workload["block_size"])
        if e and e.get("kernels") == kernel_sources_hash(what) and all(e["workload"].get(k) == v for k, v in workload.items()):
            return {"bytes_per_launch": e["bytes_per_launch"], "read": e["read"], "write": e["write"], "source": e["source"],
                    "kernels": e["kernels"], "measured_in_this_run": False}
    except Exception:
        pass
    return None


def measure_traffic_live(level, bs, n_blocks, algo_bytes):
    """HBM bytes per launch measured IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE: they do not
    fit one pass, MI355X_MICROARCH.md) over a child process that decodes a 10-tile corpus with the same library on this box; the decode
    kernels' counters of the timed launches, corrected as tools/profile_summary.py does (FETCH_SIZE counts requests x 64 B: x 1.107 for
    this kernel's single-sector gathers, profiles/r3_gather_calibration.log; WRITE_SIZE as counted), per block x this run's blocks
    (traffic per block does not depend on the launch's size beyond the caches: 10 tiles = 2.1 GB decoded >> 256 MiB Infinity Cache).
    -> dict, or None when rocprofv3 is not there / a pass fails (the caller then falls back to the committed table)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("ZXC_BENCH_LIVE_TRAFFIC", "1") == "0" or not shutil.which("rocprofv3"):
        return None
    kernels = ("zxc_decode_blocks_lean_kernel", "zxc_decode_blocks_kernel", "zxc_decode_blocks_lean_pre_kernel", "zxc_rle_expand_kernel",
               "zxc_pivco_sections_small_kernel", "zxc_pivco_sections_medium_kernel", "zxc_pivco_sections_large_kernel")
    child_tiles, steps = 10, 3
    got, child_blocks = {}, None
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="zxc_traffic_", dir="/tmp")
            env = dict(os.environ, ZXC_BENCH_LIVE_TRAFFIC="0", TMPDIR="/tmp")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            cmd = ["rocprofv3", "--pmc", ctr, "-d", d, "-o", "t", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--tiles", str(child_tiles), "--steps", str(steps), "--warmup", "2", "--no-secondary", "--no-cpu-baseline", "--calib",
                   "--level", str(level), "--block-size", str(bs)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            line = next((json.loads(x) for x in r.stdout.splitlines() if x.startswith("{") and '"metric"' in x), None)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or line is None or not files:
                return None
            child_blocks = line["config"]["blocks_per_gpu"]
            per = {}
            for row in csv.DictReader(open(files[0])):
                k = row["Kernel_Name"].split("(")[0].strip()
                if k in kernels and row["Counter_Name"] == ctr:
                    per.setdefault(k, {}).setdefault(int(row["Dispatch_Id"]), 0.0)
                    per[k][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
            tot = 0.0
            for k, dd in per.items():  # dispatches of a kernel: [calibration, 2 warm-up, `steps` timed, 1 re-check]
                ids = sorted(dd)
                ids = ids[-(steps + 1):-1] if len(ids) >= steps + 1 else []
                if ids:
                    tot += sum(dd[i] for i in ids) / len(ids)
            got[ctr] = tot * 1024.0  # (the counters are in KiB)
            shutil.rmtree(d, ignore_errors=True)
    except Exception:
        return None
    if not got.get("FETCH_SIZE") or not got.get("WRITE_SIZE") or not child_blocks:
        return None
    scale = n_blocks / child_blocks
    read, write = got["FETCH_SIZE"] * 1.107 * scale, got["WRITE_SIZE"] * scale
    return {"bytes_per_launch": int(read + write), "read": int(read), "write": int(write), "over_algorithmic": round((read + write) / algo_bytes, 3),
            "measured_in_this_run": True, "kernels": kernel_sources_hash(),
            "how": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each) around a child process decoding a {child_tiles}-tile corpus "
                   f"({child_blocks} blocks) with this library on this box; decode kernels of the timed launches; FETCH_SIZE x 1.107 "
                   f"(single-sector gathers, profiles/r3_gather_calibration.log), per block x {n_blocks} blocks"}


def host_api_run(args, dev):
    """SURVEY.md §8(d) "also report end-to-end through the C API (incl. H2D / D2H) separately": the drop-in entry points on
    PAGEABLE host buffers, PCIe both ways inside the timed region — zxc_compress + zxc_decompress of a 1 GiB frame (five corpus
    tiles; reference entry points src/lib/zxc_dispatch.c:658-840, :842-1005) and zxc_seekable_decompress_range[_mt] over a
    reference-written archive of one tile (src/lib/zxc_seekable.c:695-785, :999-1108). Every output byte compared. Never `value`."""
    import multiprocessing as mp
    import zxc_amd
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    bs = args.block_size
    pool = mp.get_context("spawn").Pool(max(1, min(48, os.cpu_count() or 1)))
    tiles = [tile_bytes(t, bs, pool)[0] for t in range(5)]
    pool.close()
    data = b"".join(tiles)[:1 << 30]
    L = zxc_amd.lib()
    L.zxc_compress_bound.restype = C.c_uint64
    L.zxc_compress_bound.argtypes = [C.c_size_t]
    L.zxc_compress.restype = C.c_int64
    L.zxc_decompress.restype = C.c_int64

    COpts = zxc_amd.api._CompressOpts  # include/zxc_opts.h (reference include/zxc_opts.h:58-78)
    L.zxc_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(COpts)]
    L.zxc_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    o = COpts(level=3, block_size=bs, seekable=1)
    cap = int(L.zxc_compress_bound(len(data)))
    cbuf = C.create_string_buffer(cap)
    dbuf = C.create_string_buffer(len(data))
    res = {"note": "pageable host buffers in and out, H2D + D2H inside the timed region; best of 3 after one warm-up call",
           "frame_bytes": len(data), "block_size": bs}

    def best_of(fn, n=3):
        fn()
        b = None
        for _ in range(n):
            t0 = time.perf_counter()
            r = fn()
            dt = time.perf_counter() - t0
            b = dt if b is None or dt < b else b
        return r, b
    csize, dt = best_of(lambda: L.zxc_compress(data, len(data), cbuf, cap, C.byref(o)))
    assert csize > 0, csize
    res["zxc_compress"] = {"value": round(len(data) / dt / 1e9, 2), "unit": "GB/s of source", "level": 3, "ms": round(dt * 1e3, 1),
                           "ratio": round(len(data) / csize, 3)}
    comp = cbuf.raw[:csize]
    rc, dt = best_of(lambda: L.zxc_decompress(comp, csize, dbuf, len(data), None))
    assert rc == len(data) and dbuf.raw == data, "zxc_decompress: bytes differ"
    res["zxc_decompress"] = {"value": round(len(data) / dt / 1e9, 2), "unit": "GB/s decoded", "ms": round(dt * 1e3, 1),
                             "checked": "every byte == source"}
    # the push API (include/zxc_pstream.h; reference src/lib/zxc_pstream.c), fed zxc_cstream_in_size() = 128 MiB per call: every block a
    # call completes goes through the same piece pipeline; the dstream reads what the cstream wrote, every byte compared
    PS = zxc_amd.api._bind_pstream(L)
    IB, OB = zxc_amd.api._InBuf, zxc_amd.api._OutBuf
    a_data = C.cast(C.c_char_p(data), C.c_void_p).value
    state = {}

    def push_compress():
        cs = PS.zxc_cstream_create(C.byref(COpts(level=3, block_size=bs)))
        chunk = int(PS.zxc_cstream_in_size(cs))
        out = OB(C.addressof(cbuf), cap, 0)
        off = 0
        while off < len(data):
            n = min(chunk, len(data) - off)
            inb = IB(a_data + off, n, 0)
            while inb.pos < inb.size:
                assert PS.zxc_cstream_compress(cs, C.byref(out), C.byref(inb)) >= 0
            off += n
        while True:
            r = PS.zxc_cstream_end(cs, C.byref(out))
            assert r >= 0
            if r == 0:
                break
        PS.zxc_cstream_free(cs)
        state["chunk"] = chunk
        return out.pos

    def push_decompress():
        ds = PS.zxc_dstream_create(None)
        out = OB(C.addressof(dbuf), len(data), 0)
        chunk = max(1 << 16, state["chunk"] * state["csize"] // len(data))  # compressed bytes of about one window of output
        off = 0
        while not PS.zxc_dstream_finished(ds):
            inb = IB(C.addressof(cbuf) + off, min(chunk, state["csize"] - off), 0)
            r = PS.zxc_dstream_decompress(ds, C.byref(out), C.byref(inb))
            assert r >= 0 and (r > 0 or inb.pos > 0 or PS.zxc_dstream_finished(ds)), r
            off += inb.pos
        PS.zxc_dstream_free(ds)
        return out.pos
    C.memset(dbuf, 0, len(data))
    state["csize"], dt = best_of(push_compress)
    res["zxc_cstream"] = {"value": round(len(data) / dt / 1e9, 2), "unit": "GB/s of source", "level": 3, "ms": round(dt * 1e3, 1),
                          "fed_per_call_mib": state["chunk"] >> 20, "ratio": round(len(data) / state["csize"], 3)}
    rc, dt = best_of(push_decompress)
    assert rc == len(data) and dbuf.raw == data, "zxc_dstream: bytes differ"
    res["zxc_dstream"] = {"value": round(len(data) / dt / 1e9, 2), "unit": "GB/s decoded", "ms": round(dt * 1e3, 1),
                          "fed_per_call_mib": round(max(1 << 16, state["chunk"] * state["csize"] // len(data)) / 2**20, 1),
                          "checked": "every byte == source, through the archive zxc_cstream wrote"}
    del cbuf, comp
    if oracle_py.Ref.available():
        t0d = tiles[0]
        arc = oracle_py.Ref().compress(t0d, 3, bs, True, False)
        sk = zxc_amd.Seekable(arc)
        out = C.create_string_buffer(len(t0d))
        for name, nt in (("zxc_seekable_decompress_range", None), ("zxc_seekable_decompress_range_mt", 0)):
            def call():
                if nt is None:
                    return L.zxc_seekable_decompress_range(sk._h, out, len(t0d), 0, len(t0d))
                return L.zxc_seekable_decompress_range_mt(sk._h, out, len(t0d), 0, len(t0d), nt)
            rc, dt = best_of(call)
            assert rc == len(t0d) and out.raw == t0d, name
            res[name] = {"value": round(len(t0d) / dt / 1e9, 2), "unit": "GB/s decoded", "ms": round(dt * 1e3, 1),
                         "archive": f"corpus tile 0 written by the reference ({len(t0d) >> 20} MiB decoded)", "checked": "every byte"}
        sk.close()
    L.zxc_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(zxc_amd.api._DecompressOpts)]  # (as zxc_amd.api binds it)
    return res


def configs0_run(args):
    """BASELINE configs[0]: the CPU reference's own zxc_compress / zxc_decompress round trip of dickens-class text at level 3
    (10 MB; the reference's default 512 KiB blocks, and one 2 MiB block for "single block") — the stated bit-exact baseline.
    Timed on the host (reference, 1 thread); then the same archives through this library: the reference-written archive decoded
    on the GPU == the text, the archive written by zxc_compress of this library decoded by the unmodified reference == the text."""
    import zxc_amd
    from zxc_amd import corpus
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    if not oracle_py.Ref.available():
        return None
    ref = oracle_py.Ref()
    text = b"".join(corpus.gen_chunk(c) for c in corpus.enwik_chunks(16 << 20, seed=77))[:10_000_000]
    res = {"text_bytes": len(text), "level": 3, "what": "reference zxc_compress / zxc_decompress on one host core (bit-exact baseline), and the "
                                                        "same archives through libzxc_mi355x.so in both directions"}
    for name, bs, n in (("default_512k_blocks", 524288, len(text)), ("single_2m_block", 2097152, 2097152)):
        data = text[:n]
        best_c = best_d = None
        for _ in range(3):
            t0 = time.perf_counter(); arc = ref.compress(data, 3, bs, False, False); dt = time.perf_counter() - t0
            best_c = dt if best_c is None or dt < best_c else best_c
            t0 = time.perf_counter(); rc, out = ref.decompress(arc, len(data)); dt = time.perf_counter() - t0
            best_d = dt if best_d is None or dt < best_d else best_d
            assert rc == len(data) and out == data
        got = zxc_amd.decompress(arc)
        assert got == data, "GPU decode of the reference-written archive differs"
        ours = zxc_amd.compress(data, 3, bs, False)
        rc, out = ref.decompress(ours, len(data))
        assert rc == len(data) and out == data, "reference decoder rejects this library's archive"
        res[name] = {"bytes": len(data), "block_size": bs, "ratio_reference": round(len(data) / len(arc), 3), "ratio_this_library": round(len(data) / len(ours), 3),
                     "reference_compress_MBs_1t": round(len(data) / best_c / 1e6, 1), "reference_decompress_MBs_1t": round(len(data) / best_d / 1e6, 1),
                     "gpu_decode_of_reference_archive": "bit-exact", "reference_decode_of_gpu_archive": "bit-exact"}
    return res


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start N ranks of this script through torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1) and hand their exit code back — never a silent one-GPU run."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def rehearse(args, comm):
    """The N-rank path without a GPU (tests/test_sharding_cpu.py): gloo ranks on CPU tensors go through every step of
    decode_run except the device launch — partition, reference-encoded tiles of the range, all_gather of the seek-table
    entries, ONE table through zxc_seekable_open_reader, zxc_mi355x_plan_seekable — and the oracle (the checker) decodes
    each rank's jobs from its own compressed span."""
    import torch
    from zxc_amd import corpus
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    rank, world, local, backend, dist = comm
    corpus.TILE_BYTES = 5 * 65536
    corpus.CHUNK_BYTES = 2 * 65536
    bs = 65536
    total_tiles = (args.total_tiles or 8 * args.tiles) if args.scaling == "strong" else 0
    n_total, first, last = rank_partition(rank, world, args.tiles, bs, total_tiles)
    d_comp, my_sizes, d_want, hdr, eof, prep, _ = build_rank_corpus(first, last, args.level, bs, None, torch.device("cpu"))
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, my_sizes.tobytes())
        all_sizes = np.concatenate([np.frombuffer(b, dtype=np.uint32) for b in gathered])
    else:
        all_sizes = my_sizes
    s = open_global_table(all_sizes, bs, hdr, eof, n_total * bs)
    jobs = s.plan(first, last - first, 16 + int(all_sizes[:first].astype(np.int64).sum()))
    comp = d_comp.numpy().tobytes()
    o = oracle_py.Oracle()
    out = bytearray()
    for j in jobs:
        rc, b = o.decode_block(comp[int(j["comp_off"]):int(j["comp_off"]) + int(j["comp_size"])], bs)
        assert rc == int(j["out_len"])
        out += b
    assert bytes(out) == d_want.numpy().tobytes()
    t = torch.tensor([float(len(out))], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"metric": "rehearsal of the N-rank launch path on CPU tensors (no device launch, nothing timed)", "value": 0.0,
                          "unit": "GB/s", "n_gpus": world, "rehearsal": True, "scaling": args.scaling, "blocks_total": n_total, "blocks_rank0": [first, last],
                          "decoded_bytes_all_ranks": int(t.item()), "checker": "oracle decode of every rank's jobs == corpus"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tiles", type=int, default=int(os.environ.get("ZXC_BENCH_TILES", "41")),
                    help="corpus tiles per GPU, 211 943 424 B of unique silesia-mix plaintext each (41 = 8.7 GB decoded / "
                         "4.4 GB compressed per GPU = configs[3]'s 64 GiB over 8 GPUs)")
    ap.add_argument("--enc-mib", type=int, default=1024, help="MiB of unique enwik-like text per GPU for the encoder workload (configs[2]: 1 GiB)")
    ap.add_argument("--l7-tiles", type=int, default=10, help="corpus tiles of the level-7 line of the default run (configs[4])")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--block-size", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default run only: skip the level-7 decode (configs[4]) and level-3 encode (configs[2]) lines")
    ap.add_argument("--checksum", action="store_true",
                    help="archives carry per-block rapidhash trailers and every launch verifies them on the device")
    ap.add_argument("--calib", action="store_true",
                    help="(tools/profile.sh) first launch = a RAW-only archive of known size: 16 B/lane streaming reads "
                         "and writes of known byte counts in the same counter pass, to calibrate FETCH_SIZE / WRITE_SIZE")
    ap.add_argument("--mode", choices=("decode", "encode"), default="decode",
                    help="decode = the headline metric (BASELINE.json configs[1]; --level 7 gives configs[4]); "
                         "encode = configs[2]: device match finder + serialiser over enwik-like text, GB/s of source")
    ap.add_argument("--rehearse", action="store_true", help="CPU rehearsal of the N-rank launch path (gloo, no device launch)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): --tiles per GPU, the corpus grows with N; strong: ONE corpus of --total-tiles tiles (BASELINE "
                         "configs[3] as worded: 64 GiB at 1/2/4/8 GPUs), rank g decodes its block range of it")
    ap.add_argument("--total-tiles", type=int, default=0,
                    help="--scaling strong: tiles of the one corpus (default 8 x --tiles = 328 tiles = 64.7 GiB decoded; N = 1 holds "
                         "all of it: 69 GB out + 69 GB reference copy + 35 GB compressed of the 288 GB)")
    ap.add_argument("--max-prep-s", type=float, default=float(os.environ.get("ZXC_BENCH_MAX_PREP_S", "1500")),
                    help="refuse a run whose (untimed) input preparation is estimated above this many seconds per rank")
    args = ap.parse_args()
    if args.rehearse:
        os.environ.setdefault("ZXC_BENCH_BACKEND", "gloo")
    # --gpus N is the contract: N ranks, one per GPU. Under a launcher WORLD_SIZE must agree with it; without one this
    # process starts the ranks itself.
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")

    if args.rehearse:
        comm = init_ranks(cpu=True)
        rehearse(args, comm)
    else:
        import torch
        import zxc_amd
        comm = init_ranks()
        rank, world, local, backend, dist = comm
        torch.cuda.set_device(local)
        zxc_amd.lib().zxc_mi355x_set_device(local)
        if args.mode == "encode":
            line = encode_run(args, args.level, args.enc_mib, args.steps, args.warmup, comm, not args.no_cpu_baseline)
        else:
            # (the headline line of the default run measures its HBM traffic itself: two short rocprofv3 passes in a child process)
            args.live_traffic = world == 1 and args.level == 3 and not args.checksum and not args.calib and not args.no_secondary and args.block_size == 65536
            line = decode_run(args, args.level, args.tiles, args.steps, args.warmup, comm, args.checksum, args.calib,
                              not args.no_cpu_baseline)
            args.live_traffic = False
            # The default single-GPU run also measures the other two single-GPU configurations of BASELINE.json, each with its
            # own roofline / cpu_baseline objects and its own bit-exactness check (next #2).
            if world == 1 and args.level == 3 and not args.checksum and not args.calib and not args.no_secondary:
                sec = {}
                sec["level7"] = decode_run(args, 7, args.l7_tiles, max(5, args.steps // 2), 2, comm,
                                           with_cpu_baseline=not args.no_cpu_baseline, cpu_budget_s=6.0)
                sec["encode_l3"] = encode_run(args, 3, args.enc_mib, max(3, args.steps // 4), 1, comm, not args.no_cpu_baseline)
                # the encoder at the reference's default block size (include/zxc_constants.h:60 there): levels 3-5 switch to the entry with the
                # 2^15-position chain ring above 64 KiB blocks (VERDICT r5 missing #3; its size against the reference's: configs0_cpu_reference below)
                sec["encode_l3_512k"] = encode_run(args, 3, min(args.enc_mib, 256), 3, 1, comm, False, block_size=524288, extras=False)
                sec["host_api"] = host_api_run(args, torch.device("cuda", local))
                # the block sizes the reference defaults to (512 KiB; maximum 2 MiB: include/zxc_constants.h:60-64 there): the SAME corpus
                # bytes re-encoded by the reference at that block size, device-resident, one wavefront per block
                for key, bsz in (("block_512k", 524288), ("block_2m", 2097152)):
                    sec[key] = decode_run(args, 3, args.tiles, max(5, args.steps // 2), 2, comm, with_cpu_baseline=not args.no_cpu_baseline,
                                          cpu_budget_s=4.0, block_size=bsz)
                sec["configs0_cpu_reference"] = configs0_run(args)
                line["secondary"] = sec
        if rank == 0:
            print(json.dumps(line))
    if comm[4] is not None:
        comm[4].destroy_process_group()


if __name__ == "__main__":
    main()
