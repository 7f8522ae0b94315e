#!/bin/bash
# tools/kprof_enc.sh <tag> : counter passes of the encoder kernels on tools/encab.py's workload (256 MiB of text, levels 1/3/5/7)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TD_TD_BUSY_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TD_TC_STALL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $set -d $R/gpurun_out/${tag}_ep$i -o p --output-format csv -- python $R/tools/encab.py > $R/gpurun_out/${tag}_ep$i.log 2>&1
done
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_ept -o p --output-format csv -- python $R/tools/encab.py > $R/gpurun_out/${tag}_ept.log 2>&1
python - <<PY
import csv, glob, collections, os
G = "$R/gpurun_out"
for kern in ("zxc_encode_blocks_kernel_l1", "zxc_encode_blocks_kernel_l3", "zxc_encode_blocks_kernel_l57"):
    vals = {}
    for d in sorted(glob.glob(f"{G}/${tag}_ep[0-9]*")):
        f = os.path.join(d, "p_counter_collection.csv")
        if not os.path.isfile(f): continue
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(kern): per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        for c, dd in per.items(): vals[c] = sum(dd.values()) / len(dd)
    nb = vals.get("SQ_WAVES", 1)
    print(f"== {kern}: waves (= blocks) per launch {nb:.0f}")
    for c in sorted(vals): print(f"  {c:34s} {vals[c]:16.0f}   per block {vals[c]/nb:12.1f}")
for r in csv.DictReader(open(f"{G}/${tag}_ept/p_kernel_stats.csv")):
    if "encode" in r["Name"]: print(f"  kernel {r['Name'][:40]:40s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:10.1f} us")
PY
