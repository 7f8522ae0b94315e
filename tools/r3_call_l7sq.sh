#!/bin/bash
# SQ issue / wait counters of the level-7 launch (full kernel), plus an EXP_PIV_PROF clock split when that variant exists.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export AB_LEVEL=7 AB_TILES=4
cd $R && python tools/abbench.py > $R/gpurun_out/r3s_ab.log 2>&1
tail -3 $R/gpurun_out/r3s_ab.log
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum TA_BUSY_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set -d $R/gpurun_out/r3s_kp$i -o p --output-format csv -- python $R/tools/abbench.py --one > $R/gpurun_out/r3s_kp$i.log 2>&1
done
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3s_kpt -o p --output-format csv -- python $R/tools/abbench.py --one > $R/gpurun_out/r3s_kpt.log 2>&1
python $R/tools/kprof_summary.py r3s zxc_decode_blocks_kernel | tee $R/gpurun_out/r3s_summary.txt
