#!/usr/bin/env python3
"""Condense the rocprofv3 CSV outputs of tools/profile.sh (gpurun_out/<tag>_{kt,fetch,write,sq,tcp}) into
profiles/<tag>_summary.json + <tag>_kernel_stats.csv, and enter the launch's HBM traffic into profiles/r6_traffic.json
(the table bench.py's roofline.traffic is read from) together with the content hash of the device sources that were profiled
(bench.kernel_sources_hash(): bench.py reports the traffic only while that hash still matches its own tree).

usage: profile_summary.py <tag> [decode|encode]

Round 3: a decode launch is several kernels (zxc_hip_shim.hip): zxc_decode_blocks_lean_kernel over every block beside
zxc_decode_blocks_kernel over the list of blocks only it can decode; at levels 6-7 also the three workgroup section kernels
(zxc_pivco_sections_*) and the lean kernel's second entry over the blocks they prepared. Counters are summed over all of
them per launch, the launch time is their span in the kernel trace. FETCH_SIZE is corrected per access pattern (profiles/r3_gather_calibration.log): the
calibration launch of the run (RAW blocks = a 16 B/lane stream of known size) gives the STREAM factor (x1.98: 128-byte
requests tallied at 64 B); the decode kernels' own reads are single-sector gathers and short runs, for which the counter
reads x1.0 .. x1.2 of the truth: the raw counter x1.107 (the 16-byte gather figure) is reported, with the bracket."""
import collections, csv, json, os, shutil, sys
tag = sys.argv[1]
what = sys.argv[2] if len(sys.argv) > 2 else "decode"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles"); os.makedirs(P, exist_ok=True)
DEC = ("zxc_decode_blocks_lean_kernel", "zxc_decode_blocks_kernel", "zxc_decode_blocks_lean_pre_kernel", "zxc_rle_expand_kernel",
       "zxc_pivco_sections_small_kernel", "zxc_pivco_sections_medium_kernel", "zxc_pivco_sections_large_kernel")
GATHER_FACTOR, GATHER_BRACKET = 1.107, (1.0, 1.2)


def is_ours(name):
    if what == "encode":
        return name.startswith("zxc_encode_blocks_kernel")
    return any(name.split("(")[0].strip() == k for k in DEC)


def counters(path, skip_first):
    """-> {counter: mean per launch}, launches. Per kernel the dispatches are taken in order; the first one of each decode
    kernel is the calibration launch (bench.py --calib) and is returned separately."""
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].strip()
        if is_ours(k):
            per[r["Counter_Name"]][k][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    mean, calib, n = {}, {}, 0
    for c, bykern in per.items():
        tot = 0.0
        for k, d in bykern.items():
            ids = sorted(d)
            if skip_first and ids:
                # bench.py --calib --warmup 2 --steps 5 launches [calibration, 2 warm-up, 5 timed, 1 re-check]; a kernel that is
                # not part of every launch (the section kernels: the launch plan follows the previous launch) is still in the
                # five timed ones if in any: the five dispatches in front of its last
                calib[c] = calib.get(c, 0.0) + d[ids[0]]
                ids = ids[-6:-1] if len(ids) >= 6 else []
            if what == "encode":  # bench.py --mode encode --warmup 1 --steps 5: [warm-up, 5 timed] over the text; the launches behind them are its incompressible-input leg
                ids = ids[1:6]
            if ids:
                tot += sum(d[i] for i in ids) / len(ids)
                n = max(n, len(ids))
        mean[c] = tot
    return mean, calib, n


out = {"tag": tag, "what": what, "kernels": list(DEC) if what == "decode" else "zxc_encode_blocks_kernel_*"}
ks = os.path.join(G, f"{tag}_kt", "kt_kernel_stats.csv")
if os.path.exists(ks):
    shutil.copyfile(ks, os.path.join(P, f"{tag}_kernel_stats.csv"))
    out["kernel_stats"] = [{"name": r["Name"][:60], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 1),
                            "pct": float(r["Percentage"])} for r in csv.DictReader(open(ks)) if is_ours(r["Name"].split("(")[0].strip())]
kt = os.path.join(G, f"{tag}_kt", "kt_kernel_trace.csv")
calib_launch = what == "decode"
if os.path.exists(kt):
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].strip())
                  for r in csv.DictReader(open(kt)) if is_ours(r["Kernel_Name"].split("(")[0].strip()))
    # group the kernels of one launch. Decode: every launch has exactly one zxc_decode_blocks_lean_kernel dispatch, started right
    # behind the launch-order pass; a dispatch belongs to the last launch whose lean kernel started no later than 100 us after it.
    # Encode: one kernel per launch.
    launches = []
    if what == "decode":
        starts = sorted(s for s, e, k in rows if k == "zxc_decode_blocks_lean_kernel")
        spans = [[None, None, []] for _ in starts]
        for s, e, k in rows:
            j = max((i for i, t in enumerate(starts) if t <= s + 100000), default=0)
            sp = spans[j]
            sp[0] = s if sp[0] is None else min(sp[0], s)
            sp[1] = e if sp[1] is None else max(sp[1], e)
            sp[2].append(k)
        launches = [tuple(sp) for sp in spans if sp[0] is not None]
    else:
        launches = [(s, e, [k]) for s, e, k in rows]
    durs = [e - s for s, e, _ in launches]
    # bench.py --calib --warmup 2 --steps 5: [calibration, 2 warm-up, 5 timed, 1 re-check]; encode: 1 warm-up, 5 timed, verification decode
    timed = durs[3:8] if calib_launch else durs[1:6]
    out["kernel_trace"] = {"launches_seen": len(durs), "steady_state_avg_ns": sum(timed) / max(1, len(timed)), "steady_state_launches": len(timed),
                           "launch = ": "span of the launch's kernels in the trace (lean + full kernel run side by side)" if what == "decode" else "the encode kernel"}
log = os.path.join(G, f"{tag}_kt.log")
bench = None
if os.path.exists(log):
    for line in open(log):
        if line.startswith("{") and '"metric"' in line:
            bench = json.loads(line)
            out["bench_line"] = {k: bench[k] for k in ("metric", "value", "ms_per_step", "roofline", "calibration") if k in bench}
            out["bench_line"]["workload"] = bench["config"]["workload"]
pm, cal = {}, {}
for sub, f in (("fetch", "f"), ("write", "w"), ("sq", "s"), ("tcp", "t")):
    p = os.path.join(G, f"{tag}_{sub}", f"{f}_counter_collection.csv")
    if os.path.exists(p):
        m, c, n = counters(p, calib_launch)
        pm.update(m); cal.update(c)
        out.setdefault("launches_averaged", {})[sub] = n
out["pmc_mean_per_launch"] = {k: round(v, 1) for k, v in pm.items()}
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    raw_r, raw_w = pm["FETCH_SIZE"] * 1024, pm["WRITE_SIZE"] * 1024
    c = (bench or {}).get("calibration")
    if c and cal.get("FETCH_SIZE"):
        out["calibration_launch"] = {**c, "FETCH_SIZE_KiB": cal["FETCH_SIZE"], "WRITE_SIZE_KiB": cal.get("WRITE_SIZE"),
                                     "stream_read_factor": round(c["read_bytes"] / (cal["FETCH_SIZE"] * 1024), 4),
                                     "write_factor": round(c["write_bytes"] / (cal["WRITE_SIZE"] * 1024), 4) if cal.get("WRITE_SIZE") else None}
    read = raw_r * GATHER_FACTOR
    out["hbm_traffic_bytes_per_launch"] = {"read": int(read), "write": int(raw_w), "total": int(read + raw_w), "raw_read": int(raw_r),
                                           "read_bracket": [int(raw_r * GATHER_BRACKET[0]), int(raw_r * GATHER_BRACKET[1])],
                                           "note": "FETCH_SIZE x 1.107 (64-byte-sector gathers and short runs: profiles/r3_gather_calibration.log; bracket x1.0 .. x1.2), "
                                                   "WRITE_SIZE as counted (exact on the calibration launch)"}
    if bench:
        algo = bench["roofline"]["algorithmic_bytes_per_launch"]
        out["hbm_traffic_bytes_per_launch"]["over_algorithmic"] = round((read + raw_w) / algo, 3)
        # the table bench.py reads
        tp = os.path.join(P, "r6_traffic.json")
        tab = json.load(open(tp)) if os.path.exists(tp) else {}
        cfg = bench["config"]
        if what == "decode":
            lvl = int(bench["metric"].split("level ")[1].split(",")[0])
            bsz = 1 << (int(cfg["decoded_bytes_per_gpu"] // cfg["blocks_per_gpu"]) - 1).bit_length()  # (the last block of a tile may be short)
            key, wl = f"decode_l{lvl}_bs{bsz}", {"tiles": cfg["prep"]["tiles"], "block_size": bsz}
        else:
            lvl = int(bench["metric"].split("level ")[1].split(",")[0])
            bsz = int(bench["metric"].split(" KiB blocks")[0].split(", ")[-1]) << 10
            key, wl = f"encode_l{lvl}_bs{bsz}", {"enc_mib": int(cfg["workload"].split(": ")[1].split(" MiB")[0]), "block_size": bsz}
        sys.path.insert(0, ROOT)
        import bench as _bench
        tab[key] = {"workload": wl, "bytes_per_launch": int(read + raw_w), "read": int(read), "write": int(raw_w),
                    "kernels": _bench.kernel_sources_hash(what),
                    "source": f"profiles/{tag}_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"}
        json.dump(tab, open(tp, "w"), indent=1)
if "TCP_TCC_READ_REQ_sum" in pm:
    out["l1_to_l2_requests_per_launch"] = {"read": int(pm["TCP_TCC_READ_REQ_sum"]), "write": int(pm.get("TCP_TCC_WRITE_REQ_sum", 0)),
                                           "note": "TCP_TCC_READ_REQ / WRITE_REQ summed over the launch's kernels (round 4: removing every far read — 1 760 of "
                                                   "the ~2 435 read requests per level-3 block — is worth 14 % of the launch: profiles/r4e_ring_size_and_no_readback.log)"}
json.dump(out, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
