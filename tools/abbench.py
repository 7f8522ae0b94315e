#!/usr/bin/env python3
"""A/B of library variants on the bench workload (GPU box only): python tools/abbench.py libA.so libB.so ...
The parent prepares AB_TILES corpus tiles once (bench.build_rank_corpus on the CPU, saved under /tmp); one child
per variant loads them, times the decode launch (best of 3 x 5) and checks every byte.
Env: AB_TILES (3), AB_LEVEL (3)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TMP = "/tmp/zxc_abbench"
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np, torch, zxc_amd
    level = int(os.environ.get("AB_LEVEL", "3"))
    BS = int(os.environ.get("AB_BLOCK_SIZE", "65536"))  # (the corpus re-encoded by the reference at this block size)
    dev = torch.device("cuda", 0)
    comp = np.load(f"{TMP}/comp.npy"); sizes = np.load(f"{TMP}/sizes.npy"); want = np.load(f"{TMP}/want.npy")
    n = min(sizes.size, int(os.environ.get("AB_MAXBLOCKS", "1000000000")))  # (AB_MAXBLOCKS: the first blocks only)
    sizes = sizes[:n]; want = want[:n * BS]
    jobs = np.zeros(n, dtype=zxc_amd.api.JOB_DTYPE)
    jobs["comp_size"] = sizes
    jobs["comp_off"] = np.concatenate([[0], np.cumsum(sizes.astype(np.uint64))[:-1]])
    jobs["out_off"] = np.arange(n, dtype=np.uint64) * BS
    jobs["out_len"] = BS
    d_comp = torch.from_numpy(comp).to(dev); d_want = torch.from_numpy(want).to(dev)
    d_jobs = torch.frombuffer(bytearray(jobs.tobytes()), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(n * BS + 256, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    def step(): zxc_amd.decode_blocks_device(d_comp.data_ptr(), d_jobs.data_ptr(), n, d_out.data_ptr(), d_st.data_ptr(), BS, False, stream)
    step(); step(); torch.cuda.synchronize()
    if os.environ.get("AB_PIVPROF"):  # library built with -DEXP_PIV_PROF: clock split of the PivCo section decoder in output bytes 64..127
        pr = d_out[:n * 65536].view(-1, 65536)[:, 64:128].cpu().numpy().view(np.uint32).astype(np.float64)
        pr = pr[(pr[:, :8] < 5e7).all(axis=1) & (pr[:, 6] > 0)]
        names = ["group set-up + small nodes", "medium nodes (work items)", "big flat: code table", "big flat: unpack", "big bitmap: set-up",
                 "big bitmap: words", "header + tree + pass 1", "level end (store drain)"]
        names[1] = ""; names[6] = "sequence offsets"
        for i, nm in enumerate(names):
            if nm: print(f"  {nm:28s} mean {pr[:, i].mean():9.0f} clk/block")
        for i, nm in ((12, "header + code lengths"), (13, "levels + flat roots"), (14, "pass 1 (sizes, popcounts)")): print(f"  {nm:28s} mean {pr[:, i].mean():9.0f} clk/block")
        print(f"  per block: big flat nodes {pr[:, 8].mean():.1f} ({pr[:, 9].mean():.0f} steps), big bitmap nodes {pr[:, 10].mean():.1f} "
              f"({pr[:, 11].mean():.0f} words), medium nodes {0:.1f}; blocks {pr.shape[0]}", flush=True)
        sys.exit(0)
    if os.environ.get("AB_PHASES") == "lean":  # lean kernel built with -DEXP_PHASES: shader clocks per phase in each block's first 64 bytes
        ph = d_out[:n * 65536].view(-1, 65536)[:, :64].cpu().numpy().view(np.uint32).astype(np.float64)
        ph = ph[(ph < 5e7).all(axis=1) & (ph[:, 1] > 0)]  # (RAW blocks and blocks of the full kernel: their bytes are data)
        names = ["loop edge", "tokens + varints", "scans + bounds", "tile cut + errors", "next tokens requested", "ring zeroing / giant", "literal + far requests",
                 "dependency analysis", "classification", "literal wait + puts", "far wait + puts", "rounds: header + near steps", "rounds: byte loops + whole-wave copies",
                 "rounds: tail + sparse finish", "final wait + flush", ""]
        tot = ph.sum()
        for i, nm in enumerate(names):
            if nm: print(f"  {nm:40s} {100 * ph[:, i].sum() / tot:5.1f} %   mean {ph[:, i].mean():9.0f} clk/block")
        print(f"  total mean {ph.sum(axis=1).mean():.0f} clk/block over {ph.shape[0]} blocks", flush=True)
        sys.exit(0)
    if os.environ.get("AB_PHASES"):  # library built with -DEXP_PHASES: per-phase shader clocks in each block's first 32 bytes
        ph = d_out[:n * 65536].view(-1, 65536)[:, :64].cpu().numpy().view(np.uint32).astype(np.float64)
        ph = ph[(ph < 5e7).all(axis=1)]  # RAW blocks never reach the sequence loop: their bytes are data
        names = ["varints: fast path / none", "long literals", "literal wait + far requests", "rounds: 16-byte group steps", "rounds: sparse finish", "flush", "loop tail",
                 "literal groups / giant", "varints: general path", "scans + bounds", "next tokens requested", "", "", "deps (registers only)", "rounds: byte loops", "rounds: whole-wave copies"]
        cnt = ph[:, 11:13].copy(); ph[:, 11:13] = 0; ph = ph[:, :16]
        tot = ph.sum()
        for i, nm in enumerate(names):
            if nm: print(f"  {nm:26s} {100 * ph[:, i].sum() / tot:5.1f} %   mean {ph[:, i].mean():9.0f} clk/block")
        print(f"  total mean {ph.sum(axis=1).mean():.0f} clk/block over {ph.shape[0]} blocks; batches/block {cnt[:, 0].mean():.1f}, "
              f"rounds/block {cnt[:, 1].mean():.1f}", flush=True)
        sys.exit(0)
    if os.environ.get("AB_TIMES"):  # library built with -DEXP_TIMES: status = start (hi 16) | duration (lo 16), units of 0.32 us
        st = d_st.cpu().numpy().view(np.uint32)
        start = (st >> 16).astype(np.int64); dur = (st & 0xFFFF).astype(np.float64) * 0.32
        piv = int(np.median(start)); start = (((start - piv + 0x8000) & 0xFFFF) - 0x8000).astype(np.float64) * 0.32
        start -= start.min(); end = start + dur
        print(f"blocks {st.size}  kernel span {end.max():.0f} us  mean block {dur.mean():.0f} us  p50 {np.percentile(dur,50):.0f}  p90 {np.percentile(dur,90):.0f}  "
              f"p99 {np.percentile(dur,99):.0f}  p99.9 {np.percentile(dur,99.9):.0f}  max {dur.max():.0f}")
        print(f"sum(block time)/span = average residency {dur.sum()/end.max():.0f} blocks ({dur.sum()/end.max()/256:.1f} per CU)")
        T = end.max(); bins = 20; edges = np.linspace(0, T, bins + 1)
        res = [(np.minimum(end, edges[i+1]) - np.maximum(start, edges[i])).clip(0).sum() / (edges[i+1]-edges[i]) for i in range(bins)]
        print("residency per time bin:", " ".join(f"{r:.0f}" for r in res))
        worst = np.argsort(-dur)[:12]
        print("slowest blocks (index, us, compressed size):", [(int(i), int(dur[i]), int(sizes[i])) for i in worst])
        slots = 256 * 24
        print(f"sum of block times / {slots} wavefront slots = {dur.sum() / slots:.0f} us; longest block {dur.max():.0f} us; launch span {end.max():.0f} us "
              f"({'fewer blocks than slots: the launch lasts as long as its longest block' if st.size <= slots else 'rounds of wavefronts'})", flush=True)
        sys.exit(0)
    ok = bool((d_st == BS).all().item()) and torch.equal(d_out[:n * BS], d_want)
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): step()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    ok = ok and torch.equal(d_out[:n * BS], d_want)
    print(f"{os.environ.get('ZXC_LIB_VARIANT','libzxc_mi355x.so'):28s} L{level} blocks {n} ok={ok} {best:7.3f} ms {n*BS/best/1e6:8.1f} GB/s", flush=True)
elif __name__ == "__main__":
    import multiprocessing as mp
    import numpy as np, torch, bench
    from zxc_amd import corpus
    tiles = int(os.environ.get("AB_TILES", "3")); level = int(os.environ.get("AB_LEVEL", "3"))
    os.makedirs(TMP, exist_ok=True)
    with mp.get_context("spawn").Pool(min(32, os.cpu_count() or 1)) as pool:
        BS = int(os.environ.get("AB_BLOCK_SIZE", "65536"))
        d_comp, sizes, d_want, *_ = bench.build_rank_corpus(0, tiles * (corpus.TILE_BYTES // BS), level, BS, pool, torch.device("cpu"))
    np.save(f"{TMP}/comp.npy", d_comp.numpy()); np.save(f"{TMP}/sizes.npy", sizes); np.save(f"{TMP}/want.npy", d_want.numpy())
    for lib in sys.argv[1:]:
        env = dict(os.environ); env["ZXC_LIB_VARIANT"] = lib; env["ZXC_TOOLS_AB"] = "1"
        try:  # (AB_TIMEOUT: a hung variant must not take the GPU call with it)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, timeout=float(os.environ.get("AB_TIMEOUT", "240")))
        except subprocess.TimeoutExpired:
            print(f"{lib:28s} TIMEOUT", flush=True)
