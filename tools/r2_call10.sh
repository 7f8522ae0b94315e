#!/bin/bash
# round-2 GPU call 10: PivCo decoder with preloaded small-node masks + LDS odd-depth buffer for small sections
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2k_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2k_pytest.log
timeout 300 python bench.py --level 7 --tiles 4 --steps 5 --no-cpu-baseline > gpurun_out/r2k_bench_l7.log 2>&1
timeout 300 python bench.py --level 6 --tiles 4 --steps 5 --no-cpu-baseline > gpurun_out/r2k_bench_l6.log 2>&1
tail -3 gpurun_out/r2k_pytest.log; for f in l7 l6; do tail -1 gpurun_out/r2k_bench_$f.log | cut -c1-200; done
