cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && AB_TILES=10 python tools/abbench.py > /dev/null 2>&1
cd /tmp
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TD_TC_STALL_sum TD_SPI_STALL_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  ZXC_LIB_VARIANT=libzxc_al6.so timeout -k 5 120 rocprofv3 --pmc $set -d $R/gpurun_out/r3d_kp$i -o p --output-format csv -- python $R/tools/abbench.py --one > $R/gpurun_out/r3d_kp$i.log 2>&1
done
python $R/tools/kprof_summary.py r3d zxc_decode_blocks_lean_kernel
