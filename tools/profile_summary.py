#!/usr/bin/env python3
"""Condense rocprofv3 CSV outputs (gpurun_out/<tag>_{kt,fetch,write,sq}) into profiles/<tag>_*.
Usage: python tools/profile_summary.py r1"""
import collections, csv, json, os, shutil, sys
tag = sys.argv[1]
KERNEL_ARG = sys.argv[2] if len(sys.argv) > 2 else None  # e.g. zxc_encode_blocks_kernel_l34 for the encode bench
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles"); os.makedirs(P, exist_ok=True)
KERNEL = KERNEL_ARG or "zxc_decode_blocks_kernel"

CALIB = {}  # counter -> value of the calibration launch (bench.py --calib: the first decode-kernel dispatch)


def counters(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if KERNEL in r["Kernel_Name"]:
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    if KERNEL_ARG is None:
        for c, d in per.items():  # bench.py --calib: first launch = RAW-only archive of known size, not a workload launch
            first = min(d)
            CALIB[c] = d.pop(first)
    return {c: sum(d.values()) / len(d) for c, d in per.items()}, {c: len(d) for c, d in per.items()}

out = {"kernel": KERNEL, "tag": tag}
ks = os.path.join(G, f"{tag}_kt", "kt_kernel_stats.csv")
shutil.copyfile(ks, os.path.join(P, f"{tag}_kernel_stats.csv"))
for r in csv.DictReader(open(ks)):
    if KERNEL in r["Name"]:
        out["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                               "max_ns": float(r["MaxNs"]), "pct_of_gpu_time": float(r["Percentage"])}
# steady state only: bench.py --calib --warmup 2 --steps 5 launches [calibration, 2 warm-up, 5 timed, 1 re-check]
kt = os.path.join(G, f"{tag}_kt", "kt_kernel_trace.csv")
if os.path.exists(kt):
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
               for r in csv.DictReader(open(kt)) if KERNEL in r["Kernel_Name"])
    timed = [x[1] for x in (d[3:8] if KERNEL_ARG is None else d[1:6])]  # encode bench: 1 warm-up, 5 timed
    if timed:
        out["kernel_trace"]["steady_state_avg_ns"] = sum(timed) / len(timed)
        out["kernel_trace"]["steady_state_launches"] = len(timed)
        if KERNEL_ARG is None:
            out["kernel_trace"]["calibration_launch_ns"] = d[0][1]
log = os.path.join(G, f"{tag}_kt.log")
if os.path.exists(log):
    for line in open(log):
        if line.startswith("{") and '"metric"' in line:
            b = json.loads(line)
            out["bench_line"] = {k: b[k] for k in ("value", "ms_per_step", "config", "roofline", "calibration") if k in b}
            if "prep" in b["config"]:
                out["workload"] = {"tiles": b["config"]["prep"]["tiles"], "level": int(b["metric"].split("level ")[1].split(",")[0]),
                                   "block_size": b["config"]["decoded_bytes_per_gpu"] // b["config"]["blocks_per_gpu"]}
pm = {}
for sub, f in (("fetch", "f"), ("write", "w"), ("sq", "s")):
    p = os.path.join(G, f"{tag}_{sub}", f"{f}_counter_collection.csv")
    if os.path.exists(p):
        c, n = counters(p)
        pm.update(c)
        out.setdefault("dispatches_averaged", {}).update(n)
out["pmc_mean_per_launch"] = {k: round(v, 1) for k, v in pm.items()}
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB. MI355X_MICROARCH.md notes FETCH_SIZE can read
    # half of a wide coalesced stream on gfx950; this kernel's reads are narrow/gathered, so the raw
    # value is reported and flagged uncalibrated.
    cal = out.get("bench_line", {}).get("calibration")
    fr = fw = 1.0
    if cal and CALIB.get("FETCH_SIZE") and CALIB.get("WRITE_SIZE"):
        # known bytes of the calibration launch / what the counters said for it (KiB units)
        fr = cal["read_bytes"] / (CALIB["FETCH_SIZE"] * 1024)
        fw = cal["write_bytes"] / (CALIB["WRITE_SIZE"] * 1024)
        out["calibration"] = {"fetch_factor": round(fr, 4), "write_factor": round(fw, 4), **cal,
                              "FETCH_SIZE_KiB": CALIB["FETCH_SIZE"], "WRITE_SIZE_KiB": CALIB["WRITE_SIZE"]}
    out["hbm_traffic_bytes_per_launch"] = {"read": int(pm["FETCH_SIZE"] * 1024 * fr), "write": int(pm["WRITE_SIZE"] * 1024 * fw),
                                           "total": int(pm["FETCH_SIZE"] * 1024 * fr + pm["WRITE_SIZE"] * 1024 * fw),
                                           "raw_read": int(pm["FETCH_SIZE"] * 1024), "raw_write": int(pm["WRITE_SIZE"] * 1024),
                                           "note": "counters scaled by the factors of the calibration launch (16 B/lane streaming "
                                                   "copy of known size in the same pass)" if cal else "uncalibrated"}
json.dump(out, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
