#!/bin/bash
# round-2 GPU call 7: the default bench (41 tiles per GPU) and its profile passes
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time timeout 600 python bench.py ) > gpurun_out/r2h_bench_n1.log 2>&1
PROFILE_TIMEOUT=400 bash tools/profile.sh r2 > gpurun_out/r2_profile.log 2>&1
ZXC_BENCH_BACKEND=gloo ZXC_BENCH_DEVICE=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --tiles 6 --steps 5 > gpurun_out/r2h_bench_gloo2.log 2>&1
grep -v "^$" gpurun_out/r2h_bench_n1.log | tail -5 | cut -c1-400; tail -2 gpurun_out/r2h_bench_gloo2.log | cut -c1-300
