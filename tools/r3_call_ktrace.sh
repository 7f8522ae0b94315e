#!/bin/bash
# kernel-trace stats of one abbench child per level: tools/r3_call_ktrace.sh "7 6 3"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in $1; do
  export AB_LEVEL=$L AB_TILES=10
  (cd $R && python tools/abbench.py > /dev/null 2>&1)
  timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3y_kt$L -o p --output-format csv -- python $R/tools/abbench.py --one > $R/gpurun_out/r3y_kt$L.log 2>&1
  grep GB/s $R/gpurun_out/r3y_kt$L.log
  python - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/r3y_kt$L/p_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.3: print(f"  L$L {r['Name'][:44]:44s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):6.2f} %")
PY
done
