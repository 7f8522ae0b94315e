#!/bin/bash
# round-2 GPU call 20: two-pass launches (lean kernel at 6 waves per SIMD + full kernel over the PivCo blocks): parity, default bench, levels 6 / 7
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2w_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2w_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2w_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r2w_smoke.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2w_bench_n1.log 2>&1
timeout 300 python bench.py --level 7 --tiles 4 --steps 5 --no-cpu-baseline > gpurun_out/r2w_bench_l7.log 2>&1
timeout 300 python bench.py --level 6 --tiles 4 --steps 5 --no-cpu-baseline > gpurun_out/r2w_bench_l6.log 2>&1
tail -3 gpurun_out/r2w_pytest.log; tail -2 gpurun_out/r2w_smoke.log; for f in n1 l7 l6; do tail -1 gpurun_out/r2w_bench_$f.log | cut -c1-220; done
