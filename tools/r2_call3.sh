#!/bin/bash
# round-2 GPU call 3: parity with the top-down PivCo decoder, level 6/7 and checksummed decode benches, decode variants, encoder
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2d_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2d_pytest.log
timeout 300 python bench.py --level 7 --tiles 4 --steps 5 --no-cpu-baseline > gpurun_out/r2d_bench_l7.log 2>&1
timeout 300 python bench.py --level 6 --tiles 4 --steps 5 --no-cpu-baseline > gpurun_out/r2d_bench_l6.log 2>&1
timeout 300 python bench.py --checksum --tiles 4 --steps 5 --no-cpu-baseline > gpurun_out/r2d_bench_ck.log 2>&1
timeout 420 python tools/abbench.py libzxc_mi355x.so libzxc_mm192.so libzxc_mm256.so libzxc_lm96.so libzxc_lm32.so libzxc_sp8.so libzxc_sp2.so libzxc_tile3k.so libzxc_mi355x.so > gpurun_out/r2d_ab.log 2>&1
timeout 300 python bench.py --mode encode --level 3 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2d_enc_l3.log 2>&1
tail -3 gpurun_out/r2d_pytest.log; for f in l7 l6 ck; do tail -1 gpurun_out/r2d_bench_$f.log | cut -c1-200; done; grep "GB/s" gpurun_out/r2d_ab.log; tail -1 gpurun_out/r2d_enc_l3.log | cut -c1-400
