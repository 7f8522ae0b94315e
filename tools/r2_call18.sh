#!/bin/bash
# round-2 GPU call 18: FILE* callers after the read / decode overlap and dictionary support in zxc_stream_compress
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_stream_api.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2t_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2t_pytest.log
tail -5 gpurun_out/r2t_pytest.log
