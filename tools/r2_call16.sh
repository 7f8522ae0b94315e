#!/bin/bash
# round-2 GPU call 16 (final): parity suite, smoke, default bench, its profile passes (r2), the level-7 profile passes (r2l7)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2r_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2r_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2r_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2r_smoke.log
( time timeout 600 python bench.py ) > gpurun_out/r2r_bench_n1.log 2>&1
PROFILE_TIMEOUT=400 bash tools/profile.sh r2 > gpurun_out/r2_profile.log 2>&1
PROFILE_TIMEOUT=300 PROFILE_BENCH_ARGS="--level 7 --tiles 4" bash tools/profile.sh r2l7 > gpurun_out/r2l7_profile.log 2>&1
timeout 300 python bench.py --level 7 --tiles 4 --steps 5 > gpurun_out/r2r_bench_l7.log 2>&1
tail -2 gpurun_out/r2r_pytest.log; tail -2 gpurun_out/r2r_smoke.log; grep -v "^$" gpurun_out/r2r_bench_n1.log | tail -5 | cut -c1-600; tail -1 gpurun_out/r2r_bench_l7.log | cut -c1-300
