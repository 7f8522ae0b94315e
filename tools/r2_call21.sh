#!/bin/bash
# round-2 GPU call 21: per-kernel times of a two-pass launch (default bench workload)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2x_kt -o kt --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline ${KT_ARGS} > $R/gpurun_out/r2x_kt.log 2>&1
head -8 $R/gpurun_out/r2x_kt/kt_kernel_stats.csv | cut -c1-200
