#!/bin/bash
# SQ issue / LDS counters of the section kernels at level 7 (one abbench child per counter set)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export AB_LEVEL=7 AB_TILES=10
(cd $R && python tools/abbench.py > /dev/null 2>&1)
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set -d $R/gpurun_out/r3q_kp$i -o p --output-format csv -- python $R/tools/abbench.py --one > $R/gpurun_out/r3q_kp$i.log 2>&1
done
for k in zxc_pivco_sections_medium_kernel zxc_pivco_sections_small_kernel; do python $R/tools/kprof_summary.py r3q $k; done | tee $R/gpurun_out/r3q_summary.txt
