#!/bin/bash
# tools/kprof.sh <tag> <lib.so> [kernel-name-substring] : hardware-counter passes of ONE library variant on the A/B workload
# (tools/abbench.py's child under rocprofv3, one --pmc pass per counter set; AB_TILES tiles prepared once by the parent).
# -> gpurun_out/<tag>_kprof.txt : per-launch means of the named kernel (default: the lean decode kernel)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; lib=$2; kern=${3:-zxc_decode_blocks_lean_kernel}
[ -f /tmp/zxc_abbench/comp.npy ] || (cd $R && python tools/abbench.py > /dev/null 2>&1)
i=0
maxsets=${KPROF_SETS:-99}
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum" \
           "TD_TD_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); [ $i -gt $maxsets ] && break
  ZXC_TOOLS_AB=1 ZXC_LIB_VARIANT=$lib timeout -k 5 120 rocprofv3 --pmc $set -d $R/gpurun_out/${tag}_kp$i -o p --output-format csv -- python $R/tools/abbench.py --one > $R/gpurun_out/${tag}_kp$i.log 2>&1
done
[ $maxsets -ge 99 ] && ZXC_TOOLS_AB=1 ZXC_LIB_VARIANT=$lib timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_kpt -o p --output-format csv -- python $R/tools/abbench.py --one > $R/gpurun_out/${tag}_kpt.log 2>&1
python $R/tools/kprof_summary.py $tag $kern > $R/gpurun_out/${tag}_kprof.txt 2>&1
cat $R/gpurun_out/${tag}_kprof.txt
