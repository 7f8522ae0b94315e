#!/bin/bash
# tools/taprof.sh <tag> : vector-memory path counters (TA / TCP / TD) of the bench kernel
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU2 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 60 rocprofv3 --pmc $set -d $R/gpurun_out/${tag}_ta$i -o s --output-format csv -- $CMD > $R/gpurun_out/${tag}_ta$i.log 2>&1
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$R/gpurun_out/${tag}_ta*/s_counter_collection.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("zxc_decode_blocks_kernel"):
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in per.items():
        print(f"{c:40s} {sum(d.values())/len(d):18.0f}  n={len(d)}")
PY
