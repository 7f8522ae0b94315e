#!/bin/bash
# round-2 GPU call 4: parity with the LDS-tiled PivCo decoder, level 6/7 benches (tiled vs top-down variant), LIT_MED sweep, encoder, host API
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2e_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2e_pytest.log
timeout 300 python bench.py --level 7 --tiles 4 --steps 5 --no-cpu-baseline > gpurun_out/r2e_bench_l7.log 2>&1
timeout 300 python bench.py --level 6 --tiles 4 --steps 5 --no-cpu-baseline > gpurun_out/r2e_bench_l6.log 2>&1
timeout 420 python tools/abbench.py libzxc_mi355x.so libzxc_lm96.so libzxc_lm128.so libzxc_lm160.so libzxc_lm224.so libzxc_lm128sp8.so libzxc_mi355x.so > gpurun_out/r2e_ab.log 2>&1
timeout 300 python bench.py --mode encode --level 3 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2e_enc_l3.log 2>&1
timeout 200 python tools/hostbench.py > gpurun_out/r2e_hostbench.log 2>&1
tail -3 gpurun_out/r2e_pytest.log; for f in l7 l6; do tail -1 gpurun_out/r2e_bench_$f.log | cut -c1-200; done; grep "GB/s" gpurun_out/r2e_ab.log; tail -1 gpurun_out/r2e_enc_l3.log | cut -c1-300; cat gpurun_out/r2e_hostbench.log | tail -4
