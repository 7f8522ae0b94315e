#!/bin/bash
# round-2 GPU call 6: parity incl. dictionary compression, encoder after the LDS-only fences, bigger launches
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2g_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2g_pytest.log
for lv in 3 1; do timeout 300 python bench.py --mode encode --level $lv --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2g_enc_l$lv.log 2>&1; done
timeout 400 python bench.py --tiles 20 --no-cpu-baseline > gpurun_out/r2g_bench_t20.log 2>&1
timeout 300 python bench.py --level 6 --tiles 4 --steps 5 --no-cpu-baseline > gpurun_out/r2g_bench_l6.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2g_smoke.log 2>&1
tail -4 gpurun_out/r2g_pytest.log; for lv in 3 1; do tail -1 gpurun_out/r2g_enc_l$lv.log | cut -c1-260; done; tail -1 gpurun_out/r2g_bench_t20.log | cut -c1-300; tail -1 gpurun_out/r2g_bench_l6.log | cut -c1-200; tail -1 gpurun_out/r2g_smoke.log
