#!/bin/bash
# round-2 GPU call 12+: phase split of the decode loop (library built with -DEXP_PHASES) next to A/B variants
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
AB_TILES=3 timeout 900 python tools/abbench.py ${AB_LIBS:-libzxc_mi355x.so} > gpurun_out/r2m_ab.log 2>&1
AB_PHASES=1 ZXC_LIB_VARIANT=libzxc_phases.so timeout 300 python tools/abbench.py --one >> gpurun_out/r2m_ab.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r2m_ab.log | tail -24
