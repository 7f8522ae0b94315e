#!/bin/bash
# round-2 GPU call 15: clock split of the PivCo section decoder (library built with -DEXP_PIV_PROF), level 6
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
AB_LEVEL=${AB_LEVEL:-6} AB_TILES=2 timeout 900 python tools/abbench.py ${AB_LIBS:-libzxc_mi355x.so} > gpurun_out/r2p_piv.log 2>&1
AB_LEVEL=${AB_LEVEL:-6} AB_PIVPROF=1 ZXC_LIB_VARIANT=libzxc_pivprof.so timeout 300 python tools/abbench.py --one >> gpurun_out/r2p_piv.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r2p_piv.log | tail -14
