#!/bin/bash
# tools/profile_extra.sh <tag> : kernel-trace summaries of the secondary configs (level-7 decode, encoder)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_l7kt -o kt --output-format csv -- python $R/bench.py --level 7 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_l7kt.log 2>&1
EB_TILES=8 timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_enckt -o kt --output-format csv -- python $R/tools/encbench.py > $R/gpurun_out/${tag}_enckt.log 2>&1
ls $R/gpurun_out/${tag}_l7kt $R/gpurun_out/${tag}_enckt
