#!/bin/bash
# round-2 GPU session 1: parity of the ds_or executor, then A/B against the round-1 kernel (libzxc_v1.so)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a_pytest.log
timeout 300 python bench.py --steps 20 > gpurun_out/r2a_bench_v2.log 2>&1
ZXC_LIB_VARIANT=libzxc_v1.so timeout 300 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r2a_bench_v1.log 2>&1
KB_MIB=8 timeout 400 python tools/kbench.py text source exe chem records image16 catalogue mixed > gpurun_out/r2a_kb_v2.log 2>&1
ZXC_LIB_VARIANT=libzxc_v1.so KB_MIB=8 timeout 400 python tools/kbench.py text source exe chem records image16 catalogue mixed > gpurun_out/r2a_kb_v1.log 2>&1
bash tools/profile.sh r2a > gpurun_out/r2a_profile.log 2>&1
tail -3 gpurun_out/r2a_pytest.log; cat gpurun_out/r2a_bench_v2.log | tail -1 | cut -c1-400; tail -1 gpurun_out/r2a_bench_v1.log | cut -c1-300
