#!/bin/bash
# round-2 GPU call 22: end-of-round verification of the committed tree (parity suite, smoke, default bench)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2z_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2z_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2z_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2z_smoke.log
timeout 600 python bench.py > gpurun_out/r2z_bench_n1.log 2>&1
tail -3 gpurun_out/r2z_pytest.log; tail -2 gpurun_out/r2z_smoke.log; tail -1 gpurun_out/r2z_bench_n1.log | cut -c1-260
