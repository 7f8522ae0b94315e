#!/bin/bash
# round-2 GPU call 19: encoder tests + bench lines at levels 5-7 after the deep levels' head table went to 2^13 (48 KiB LDS, 3 waves/CU)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encode.py -m gpu -x -q > gpurun_out/r2v_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2v_pytest.log
for lv in 5 6 7; do timeout 400 python bench.py --mode encode --level $lv --steps 3 --warmup 1 > gpurun_out/r2v_enc_l$lv.log 2>&1; done
tail -3 gpurun_out/r2v_pytest.log; for lv in 5 6 7; do tail -1 gpurun_out/r2v_enc_l$lv.log | cut -c1-330; done
