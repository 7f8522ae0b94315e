#!/bin/bash
# round-2 GPU call 1: parity (new Block API / dict goldens / all levels), the bench on the to-spec workload,
# the 2-rank rehearsal of the seek-table partition on one GPU, kernel variants A/B, profile passes.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2b_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2b_pytest.log
timeout 400 python bench.py > gpurun_out/r2b_bench_n1.log 2>&1
ZXC_BENCH_BACKEND=gloo ZXC_BENCH_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --tiles 2 --steps 5 > gpurun_out/r2b_bench_gloo2.log 2>&1
timeout 300 python tools/abbench.py libzxc_mi355x.so libzxc_farearly.so libzxc_mm64.so libzxc_mm96.so libzxc_mi355x.so > gpurun_out/r2b_ab.log 2>&1
bash tools/profile.sh r2b > gpurun_out/r2b_profile.log 2>&1
tail -4 gpurun_out/r2b_pytest.log; tail -1 gpurun_out/r2b_bench_n1.log | cut -c1-1500; tail -2 gpurun_out/r2b_bench_gloo2.log | cut -c1-600; cat gpurun_out/r2b_ab.log | tail -8
