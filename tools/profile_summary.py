#!/usr/bin/env python3
"""Condense rocprofv3 CSV outputs (gpurun_out/<tag>_{kt,fetch,write,sq}) into profiles/<tag>_*.
Usage: python tools/profile_summary.py r1"""
import collections, csv, json, os, shutil, sys
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles"); os.makedirs(P, exist_ok=True)
KERNEL = "zxc_decode_blocks_kernel"

def counters(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if KERNEL in r["Kernel_Name"]:
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {c: sum(d.values()) / len(d) for c, d in per.items()}, {c: len(d) for c, d in per.items()}

out = {"kernel": KERNEL, "tag": tag}
ks = os.path.join(G, f"{tag}_kt", "kt_kernel_stats.csv")
shutil.copyfile(ks, os.path.join(P, f"{tag}_kernel_stats.csv"))
for r in csv.DictReader(open(ks)):
    if KERNEL in r["Name"]:
        out["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                               "max_ns": float(r["MaxNs"]), "pct_of_gpu_time": float(r["Percentage"])}
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
    out["hbm_traffic_bytes_per_launch"] = {"read": int(pm["FETCH_SIZE"] * 1024), "write": int(pm["WRITE_SIZE"] * 1024),
                                           "total": int((pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024),
                                           "note": "FETCH_SIZE uncalibrated on gfx950 (may under-count wide streams by 2x)"}
json.dump(out, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
