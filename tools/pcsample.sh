#!/bin/bash
# tools/pcsample.sh <tag> : rocprofv3 PC sampling (beta) of the bench kernel -> gpurun_out/<tag>_pcs/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
CMD="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
for m in "stochastic cycles 1048576" "host_trap time 1"; do
  set -- $m
  timeout -k 5 90 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 \
      -d $R/gpurun_out/${tag}_pcs_$1 -o p --output-format csv -- $CMD > $R/gpurun_out/${tag}_pcs_$1.log 2>&1
  echo "$1 rc=$?"; ls -la $R/gpurun_out/${tag}_pcs_$1/ 2>/dev/null | head
done
