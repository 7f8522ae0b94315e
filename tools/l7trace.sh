#!/bin/bash
# tools/l7trace.sh [level] : kernel timeline of one level-6/7 decode launch (10 corpus tiles) on the GPU box -> gpurun_out/r4l7t_*.log
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
L=${1:-7}
export AB_TILES=10 AB_LEVEL=$L AB_TIMEOUT=150
timeout 300 python $R/tools/abbench.py libzxc_mi355x.so 2>&1 | grep "GB/s\|TIMEOUT"
timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/r4l7t_kt -o kt --output-format csv -- python $R/tools/abbench.py --one > $R/gpurun_out/r4l7t_kt.log 2>&1
python3 - <<'PY'
import csv, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f"{R}/gpurun_out/r4l7t_kt/kt_kernel_trace.csv")) if r["Kernel_Name"].startswith("zxc_"))
# the last launch: from the last zxc_order_hist_kernel on
i0 = max(i for i, r in enumerate(rows) if r[2] == "zxc_order_hist_kernel")
t0 = rows[i0][0]
for s, e, k in rows[i0:]:
    print(f"{k:40s} start {(s - t0) / 1e6:7.3f} ms  end {(e - t0) / 1e6:7.3f} ms  ({(e - s) / 1e6:6.3f} ms)")
PY
