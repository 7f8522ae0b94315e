#!/bin/bash
# round-2 GPU call 9: parity (in-place decode, stream-ordered PivCo-encoder scratch), encode benches at levels 6 / 7 after the 4-in-flight loops
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2j_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2j_pytest.log
for lv in 6 7 5; do timeout 400 python bench.py --mode encode --level $lv --enc-mib 256 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2j_enc_l$lv.log 2>&1; done
tail -4 gpurun_out/r2j_pytest.log; for lv in 6 7 5; do tail -1 gpurun_out/r2j_enc_l$lv.log | cut -c1-330; done
