#!/bin/bash
# tools/sqprof.sh <tag> : SQ counter passes (issue / LDS / wait breakdown) of the bench kernel -> gpurun_out/<tag>_sq*/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 5 60 rocprofv3 --pmc $set -d $R/gpurun_out/${tag}_sq$i -o s --output-format csv -- $CMD > $R/gpurun_out/${tag}_sq$i.log 2>&1
  tail -2 $R/gpurun_out/${tag}_sq$i.log | cut -c1-300
done
