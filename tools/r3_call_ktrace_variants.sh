#!/bin/bash
# kernel-trace stats of abbench children for library variants at one level: tools/r3_call_ktrace_variants.sh 7 "stop1 stop2 mi355x"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export AB_LEVEL=$1 AB_TILES=10
(cd $R && python tools/abbench.py > /dev/null 2>&1)
for V in $2; do
  ZXC_LIB_VARIANT=libzxc_$V.so timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3kv_$V -o p --output-format csv -- python $R/tools/abbench.py --one > $R/gpurun_out/r3kv_$V.log 2>&1
  grep GB/s $R/gpurun_out/r3kv_$V.log
  python - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/r3kv_$V/p_kernel_stats.csv")):
    if "zxc" in r["Name"] and float(r["Percentage"]) > 0.3: print(f"  $V {r['Name'][:44]:44s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):6.2f} %")
PY
done
