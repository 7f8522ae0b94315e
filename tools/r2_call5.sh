#!/bin/bash
# round-2 GPU call 5: literal-loop A/B, then the profile passes (decode headline, level 7, encoder)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python tools/abbench.py libzxc_mi355x.so libzxc_litloop1.so libzxc_mi355x.so libzxc_litloop1.so > gpurun_out/r2f_ab.log 2>&1
timeout 400 python bench.py > gpurun_out/r2f_bench_n1.log 2>&1
bash tools/profile.sh r2 > gpurun_out/r2_profile.log 2>&1
PROFILE_BENCH_ARGS="--level 7 --tiles 4" bash tools/profile.sh r2l7 > gpurun_out/r2l7_profile.log 2>&1
bash tools/profile_encode.sh r2enc 3 > gpurun_out/r2enc_profile.log 2>&1
timeout 300 python bench.py --level 7 --tiles 4 > gpurun_out/r2f_bench_l7.log 2>&1
timeout 300 python bench.py --mode encode > gpurun_out/r2f_bench_enc.log 2>&1
grep "GB/s" gpurun_out/r2f_ab.log; tail -1 gpurun_out/r2f_bench_n1.log | cut -c1-330
