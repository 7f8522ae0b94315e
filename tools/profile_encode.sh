#!/bin/bash
# tools/profile_encode.sh <tag> [level] : rocprofv3 passes of the encode bench (configs[2]): kernel trace + stats, then the
# HBM traffic and instruction counters each in its own --pmc run -> gpurun_out/<tag>_{kt,fetch,write,sq}/ ;
# condense with: python tools/profile_summary.py <tag> zxc_encode_blocks_kernel_l3
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; lv=${2:-3}
CMD="python $R/bench.py --mode encode --level $lv --steps 5 --warmup 1 --no-cpu-baseline"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_kt -o kt --output-format csv -- $CMD > $R/gpurun_out/${tag}_kt.log 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_fetch -o f --output-format csv -- $CMD > $R/gpurun_out/${tag}_fetch.log 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_write -o w --output-format csv -- $CMD > $R/gpurun_out/${tag}_write.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES -d $R/gpurun_out/${tag}_sq -o s --output-format csv -- $CMD > $R/gpurun_out/${tag}_sq.log 2>&1
ls $R/gpurun_out/${tag}_*/
