#!/bin/bash
# round-2 GPU call 8: parity incl. PivCo encode and .zxd helpers, encode benches at levels 6 / 7, PivCo decode in-flight A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2i_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2i_pytest.log
for lv in 6 7; do timeout 400 python bench.py --mode encode --level $lv --enc-mib 256 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2i_enc_l$lv.log 2>&1; done
for v in libzxc_mi355x.so libzxc_piv12.so libzxc_piv16.so; do ZXC_LIB_VARIANT=$v timeout 300 python bench.py --level 7 --tiles 4 --steps 5 --no-cpu-baseline > gpurun_out/r2i_l7_$v.log 2>&1; done
tail -4 gpurun_out/r2i_pytest.log; for lv in 6 7; do tail -1 gpurun_out/r2i_enc_l$lv.log | cut -c1-600; done; for v in libzxc_mi355x.so libzxc_piv12.so libzxc_piv16.so; do echo $v; tail -1 gpurun_out/r2i_l7_$v.log | cut -c1-160; done
