#!/bin/bash
# round-2 GPU call 11: where does a level-6 / level-7 launch spend its time (ablation flags of the experiment build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
ZXC_LIB_VARIANT=libzxc_exp.so KB_MIB=32 KB_REPL=8 KB_LEVELS=6,7 KB_DBG=0,1024 timeout 600 python tools/kbench.py mixed text > gpurun_out/r2l_kbench.log 2>&1
cat gpurun_out/r2l_kbench.log | tail -6
