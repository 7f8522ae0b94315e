#!/bin/bash
# round-2 GPU call 13: where do the decode kernel's waves sit (SQ wait / active split, LDS bank conflicts)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --tiles 4 --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 5 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/r2n_lds -o l --output-format csv -- $CMD > $R/gpurun_out/r2n_lds.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES -d $R/gpurun_out/r2n_sq -o s --output-format csv -- $CMD > $R/gpurun_out/r2n_sq.log 2>&1
python - <<'PY'
import csv, glob, collections, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for tag in ("r2n_lds", "r2n_sq"):
    for f in glob.glob(f"{R}/gpurun_out/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            if "zxc_decode_blocks_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
        for k in sorted(acc): print(f"{tag} {k:26s} {acc[k] / max(n[k], 1):16.0f} per launch ({n[k]} launches)")
PY
tail -2 $R/gpurun_out/r2n_lds.log
