#!/usr/bin/env python3
"""Condense the passes of tools/kprof.sh: per-launch means of one kernel's counters, and per block / per batch figures."""
import csv, glob, os, sys, collections
tag, kern = sys.argv[1], sys.argv[2]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); G = os.path.join(R, "gpurun_out")
vals = {}
for d in sorted(glob.glob(os.path.join(G, f"{tag}_kp[0-9]*"))):
    f = os.path.join(d, "p_counter_collection.csv")
    if not os.path.isfile(f): continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]: per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for c, dd in per.items(): vals[c] = sum(dd.values()) / len(dd)
t = os.path.join(G, f"{tag}_kpt", "p_kernel_stats.csv")
if os.path.isfile(t):
    for r in csv.DictReader(open(t)):
        print(f"  kernel {r['Name'][:48]:48s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']} %")
        if kern in r["Name"]: vals["_avg_ns"] = float(r["AverageNs"])
for line in open(os.path.join(G, f"{tag}_kp1.log")):
    if "blocks" in line and "GB/s" in line: print("  bench line under the profiler:", line.strip())
nb = vals.get("SQ_WAVES", 0)
print(f"kernel {kern}: waves (= blocks) per launch {nb:.0f}")
for c in sorted(vals):
    if c.startswith("_"): continue
    v = vals[c]
    print(f"  {c:34s} {v:16.0f}   per block {v/nb if nb else 0:10.1f}")
if "_avg_ns" in vals and "GRBM_GUI_ACTIVE" in vals:
    print(f"  effective clock {vals['GRBM_GUI_ACTIVE']/vals['_avg_ns']:.2f} GHz (GRBM_GUI_ACTIVE / kernel time)")
