#!/bin/bash
# round-2 GPU call 2: full parity incl. the hash-chain encoder, decode variants A/B (ring size, far reads), encode bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2c_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c_pytest.log
timeout 420 python tools/abbench.py libzxc_mi355x.so libzxc_farearly.so libzxc_nofar.so libzxc_ring8k.so libzxc_ring16k.so libzxc_ring8k_t35.so libzxc_ring16k_t35.so libzxc_mm64.so libzxc_mm96.so libzxc_mi355x.so > gpurun_out/r2c_ab.log 2>&1
for lv in 3 1 5; do timeout 300 python bench.py --mode encode --level $lv --steps 5 --warmup 1 > gpurun_out/r2c_enc_l$lv.log 2>&1; done
timeout 300 python bench.py --level 7 --tiles 4 --steps 5 --no-cpu-baseline > gpurun_out/r2c_bench_l7.log 2>&1
tail -5 gpurun_out/r2c_pytest.log; grep "GB/s" gpurun_out/r2c_ab.log; for lv in 3 1 5; do tail -1 gpurun_out/r2c_enc_l$lv.log | cut -c1-900; done; tail -1 gpurun_out/r2c_bench_l7.log | cut -c1-300
