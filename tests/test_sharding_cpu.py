"""Multi-GPU path on CPU: 2 gloo ranks partition a real seek table exactly like bench.py /
the device API would (contiguous block range per rank, no data-path collective), decode their
ranges with the checker, and only the timing/accounting reduction crosses ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zxc_amd import shard
    import zxc_amd
    import oracle_py
    comp = open(os.path.join(GOLDEN, "synth", "text_200k_l3_b4k.zxc"), "rb").read()
    s = zxc_amd.Seekable(comp)
    n = s.num_blocks
    lo, hi = shard.block_range(rank, world, n)
    jobs = s.plan(lo, hi - lo, comp_rebase=0)
    # the rank's slice is self-contained: decode it with the oracle block by block
    o = oracle_py.Oracle()
    out = bytearray()
    for j in jobs:
        rc, b = o.decode_block(comp[int(j["comp_off"]):int(j["comp_off"]) + int(j["comp_size"])], 4096)
        assert rc == int(j["out_len"])
        out += b
    mine = torch.tensor([hi - lo, len(out), int(jobs["comp_size"].sum())], dtype=torch.int64)
    dist.all_reduce(mine, op=dist.ReduceOp.SUM)          # accounting only
    t = torch.tensor([0.25 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)              # bench.py's max-over-ranks time
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, bytes(out)))
    if rank == 0:
        q.put((n, s.decompressed_size, mine.tolist(), float(t.item()), gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_block_range_partition(synth_inputs, manifest):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    n, total, sums, tmax, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sums[0] == n and sums[1] == total
    assert tmax == 0.5
    gathered.sort()
    assert gathered[0][0] == 0 and gathered[-1][1] == n
    assert all(gathered[i][1] == gathered[i + 1][0] for i in range(world - 1))  # disjoint, contiguous cover
    data = synth_inputs[manifest["synth"]["text_200k_l3_b4k"]["input"]]
    assert b"".join(g[2] for g in gathered) == data


def test_range_helpers():
    from zxc_amd import shard
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 4, 8):
            r = [shard.block_range(g, w, n) for g in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def _bench_worker(rank, world, port, q, tiles_per_gpu):
    """bench.py's own partition code on CPU tensors: rank_partition -> build_rank_corpus (reference-encoded tiles,
    only the ones the range touches) -> all_gather of the seek-table entries (control plane) -> ONE seek table
    opened through zxc_seekable_open_reader -> zxc_mi355x_plan_seekable(first, n, comp_rebase); the rank's jobs
    are then decoded with the oracle straight from its own compressed span."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import bench
    import oracle_py
    from zxc_amd import corpus
    corpus.TILE_BYTES = 5 * 65536      # small tiles for the CPU rehearsal (the GPU bench uses 3234 blocks per tile)
    corpus.CHUNK_BYTES = 2 * 65536
    bs = 65536
    n_total, first, last = bench.rank_partition(rank, world, tiles_per_gpu, bs)
    d_comp, my_sizes, d_want, hdr, eof, prep, _ = bench.build_rank_corpus(first, last, 3, bs, None, torch.device("cpu"))
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, my_sizes.tobytes())
        all_sizes = np.concatenate([np.frombuffer(b, dtype=np.uint32) for b in gathered])
    else:
        all_sizes = my_sizes
    s = bench.open_global_table(all_sizes, bs, hdr, eof, n_total * bs)
    assert s.num_blocks == n_total
    rebase = 16 + int(all_sizes[:first].astype(np.int64).sum())
    jobs = s.plan(first, last - first, rebase)
    comp = d_comp.numpy().tobytes()
    o = oracle_py.Oracle()
    out = bytearray()
    for j in jobs:
        rc, b = o.decode_block(comp[int(j["comp_off"]):int(j["comp_off"]) + int(j["comp_size"])], bs)
        assert rc == int(j["out_len"]) == bs
        out += b
    assert bytes(out) == d_want.numpy().tobytes()
    res = (rank, first, last, all_sizes.tobytes(), bytes(out))
    if world > 1:
        allres = [None] * world
        dist.all_gather_object(allres, res)
        if rank == 0:
            q.put(allres)
        dist.barrier()
        dist.destroy_process_group()
    else:
        q.put([res])


def test_bench_partition_is_one_seek_table(ref):
    """2 ranks x 2 tiles and 1 rank x 4 tiles are the SAME corpus and the SAME seek table; the two ranks'
    ranges are disjoint, contiguous, and together decode to what the single rank decodes."""
    ctx = mp.get_context("spawn")
    results = {}
    for world, tiles in ((2, 2), (1, 4)):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q, tiles)) for r in range(world)]
        for p in procs:
            p.start()
        results[world] = sorted(q.get(timeout=300))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    two, one = results[2], results[1]
    assert two[0][3] == two[1][3] == one[0][3]                       # one table, identical on every rank and at N = 1
    assert two[0][1] == 0 and two[0][2] == two[1][1] and two[1][2] == one[0][2] == 20
    assert two[0][4] + two[1][4] == one[0][4]


def _run_bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=600)
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    return r.returncode, lines, r.stderr


def test_bench_gpus_flag_starts_the_ranks_itself(ref):
    """`python bench.py --gpus 2` with no launcher around it must run TWO ranks (the flag used to be
    decorative): the CPU rehearsal of the launch path (gloo, CPU tensors, the oracle as the checker) prints n_gpus = 2
    and both ranks' ranges decode."""
    rc, lines, err = _run_bench(["--gpus", "2", "--rehearse", "--tiles", "2"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["blocks_total"] == 20
    assert lines[0]["blocks_rank0"] == [0, 10] and lines[0]["decoded_bytes_all_ranks"] == 20 * 65536
    rc1, lines1, err1 = _run_bench(["--gpus", "1", "--rehearse", "--tiles", "4"])
    assert rc1 == 0 and lines1[0]["n_gpus"] == 1 and lines1[0]["decoded_bytes_all_ranks"] == 20 * 65536, err1[-2000:]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """Under a launcher WORLD_SIZE must equal --gpus: a mismatch exits non-zero instead of printing an n_gpus = 1 line."""
    rc, lines, err = _run_bench(["--gpus", "8", "--rehearse"], env_extra={"WORLD_SIZE": "1", "RANK": "0"}, drop=())
    assert rc != 0 and not lines and "--gpus 8" in err


def test_bench_strong_scaling_is_one_fixed_corpus(ref):
    """--scaling strong (BASELINE configs[3] as worded: ONE corpus at 1/2/4/8 GPUs): the corpus has --total-tiles tiles whatever
    the world size; 2 ranks split the same 20 blocks 1 rank decodes alone; the prep-time guard refuses what it estimates too long."""
    import bench
    assert bench.corpus_tiles(8, 41) == 328 and bench.corpus_tiles(8, 41, 328) == 328 and bench.corpus_tiles(1, 41, 328) == 328
    rc, lines, err = _run_bench(["--gpus", "2", "--rehearse", "--scaling", "strong", "--total-tiles", "4", "--tiles", "1"])
    assert rc == 0, err[-2000:]
    assert lines[0]["n_gpus"] == 2 and lines[0]["scaling"] == "strong" and lines[0]["blocks_total"] == 20 and lines[0]["blocks_rank0"] == [0, 10]
    rc1, lines1, err1 = _run_bench(["--gpus", "1", "--rehearse", "--scaling", "strong", "--total-tiles", "4", "--tiles", "1"])
    assert rc1 == 0 and lines1[0]["blocks_total"] == 20 and lines1[0]["blocks_rank0"] == [0, 20], err1[-2000:]
    assert lines1[0]["decoded_bytes_all_ranks"] == lines[0]["decoded_bytes_all_ranks"] == 20 * 65536
    assert bench.prep_estimate_s(41, 3, 8) < bench.prep_estimate_s(328, 3, 1) < bench.prep_estimate_s(328, 7, 1)
