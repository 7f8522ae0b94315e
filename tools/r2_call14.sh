#!/bin/bash
# round-2 GPU call 14: encoder occupancy A/B at level 3 (head-table / chain-ring sizes of the levels 3-4 kernel)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; : > gpurun_out/r2o_enc_ab.log
for lib in ${AB_LIBS:-libzxc_mi355x.so}; do
  echo "== $lib" >> gpurun_out/r2o_enc_ab.log
  ZXC_LIB_VARIANT=$lib timeout 300 python bench.py --mode encode --level ${ENC_LEVEL:-3} --enc-mib 256 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r2o_enc_ab.log
done
python - <<'PY'
import json
for line in open("gpurun_out/r2o_enc_ab.log"):
    if line.startswith("=="): print(line.strip(), end="  ")
    elif line.startswith("{"):
        d = json.loads(line); print(d["value"], d["unit"], "ratio", d["config"].get("ratio"), d["config"].get("round_trip"))
    else: print(line.strip())
PY
