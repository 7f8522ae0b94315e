#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs: per-kernel time and PMC sums per dispatch."""
import sqlite3, sys, collections
for f in sys.argv[1:]:
    db = sqlite3.connect(f)
    print("==", f)
    for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("  kernel %-60s calls %4d total %10.0f ns avg %10.0f ns %5.1f%%" % (r[0][:60], r[1], r[2], r[3], r[4]))
    rows = db.execute("select name, dispatch_id, counter_name, sum(counter_value) from pmc_events group by name, dispatch_id, counter_name").fetchall()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, d, c, v in rows: agg[name][c].append(v)
    for name, cs in agg.items():
        print("  ", name[:70])
        for c, vals in sorted(cs.items()):
            print("      %-28s mean/dispatch %16.0f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
