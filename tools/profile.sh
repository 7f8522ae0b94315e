#!/bin/bash
# tools/profile.sh <tag> [bench args...] : rocprofv3 passes of a bench command (kernel trace + stats, then the HBM traffic
# counters and the instruction counts, each in its own --pmc run) -> gpurun_out/<tag>_{kt,fetch,write,sq}/ ; condense with
# python tools/profile_summary.py <tag> [what]
# Default command: the headline decode (python bench.py --no-secondary --calib). Examples:
#   tools/profile.sh r3              headline (configs[1])
#   tools/profile.sh r3l7 --level 7 --tiles 10          level-7 decode (configs[4])
#   tools/profile.sh r3enc --mode encode --enc-mib 1024 encoder (configs[2])
cd /tmp && export TMPDIR=/tmp
export ZXC_BENCH_CACHE=/tmp/zxc_bench_cache   # (the passes share the reference-encoded corpus)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
extra="$*"
case "$extra" in *--mode\ encode*) base="--steps 5 --warmup 1 --no-cpu-baseline";; *) base="--steps 5 --warmup 2 --no-cpu-baseline --no-secondary --calib";; esac
CMD="python $R/bench.py $base $extra"
T=${PROFILE_TIMEOUT:-400}
P=${PROFILE_PASSES:-kt fetch write sq tcp}   # (subset of the passes, e.g. PROFILE_PASSES="kt fetch write")
for pass in $P; do
  case $pass in
    kt)    timeout -k 5 $T rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_kt -o kt --output-format csv -- $CMD > $R/gpurun_out/${tag}_kt.log 2>&1;;
    fetch) timeout -k 5 $T rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_fetch -o f --output-format csv -- $CMD > $R/gpurun_out/${tag}_fetch.log 2>&1;;
    write) timeout -k 5 $T rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_write -o w --output-format csv -- $CMD > $R/gpurun_out/${tag}_write.log 2>&1;;
    sq)    timeout -k 5 $T rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES -d $R/gpurun_out/${tag}_sq -o s --output-format csv -- $CMD > $R/gpurun_out/${tag}_sq.log 2>&1;;
    tcp)   timeout -k 5 $T rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/${tag}_tcp -o t --output-format csv -- $CMD > $R/gpurun_out/${tag}_tcp.log 2>&1;;
  esac
done
ls $R/gpurun_out/${tag}_*/ | head -30
