#!/bin/bash
# round-2 GPU call 17: encoder main loop with / without its emission stores (raw timing, tools/encbench.py)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; : > gpurun_out/r2s_enc.log
for lib in ${AB_LIBS:-libzxc_mi355x.so}; do
  echo "== $lib" >> gpurun_out/r2s_enc.log
  ZXC_LIB_VARIANT=$lib EB_MIB=64 EB_TILES=4 timeout 300 python tools/encbench.py 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/r2s_enc.log
done
cat gpurun_out/r2s_enc.log
