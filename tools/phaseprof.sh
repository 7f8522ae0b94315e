#!/bin/bash
# tools/phaseprof.sh : dynamic instruction counts of the lean kernel's phases (GPU box). One SQ --pmc pass over the shipped library and over
# the timing-only ablation builds (ABL_*: wrong output, a phase removed): the differences are what each phase issues per block.
# Needs zxc_amd/libzxc_{base,nodeps,nonear,nolit,nofar}.so (tools/build_variant.sh <name> -DABL_...).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export AB_TILES=${AB_TILES:-10} AB_TIMEOUT=100
for v in base nodeps nonear nolit nofar; do
  ZXC_TOOLS_AB=1 ZXC_LIB_VARIANT=libzxc_$v.so timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $R/gpurun_out/r5p_sq_$v -o s --output-format csv -- python $R/tools/abbench.py --one > $R/gpurun_out/r5p_sq_$v.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
res = {}
for v in ("base", "nodeps", "nonear", "nolit", "nofar"):
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(f"{R}/gpurun_out/r5p_sq_{v}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Kernel_Name", "").startswith("zxc_decode_blocks_lean_kernel"):
                tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    res[v] = {k: tot[k] / max(n[k], 1) / 32340 for k in tot}
keys = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
print("per level-3 block (32 340 blocks, ~53 batches each)      " + "  ".join(f"{k[9:]:>8s}" for k in keys))
for v in res: print(f"{v:54s} " + "  ".join(f"{res[v].get(k, 0):8.0f}" for k in keys))
b = res.get("base", {})
for name, v in (("dependency analysis (base - nodeps)", "nodeps"), ("near rounds + dependency analysis (base - nonear)", "nonear"),
                ("literal loads + puts (base - nolit)", "nolit"), ("far puts (base - nofar; the far loads stay)", "nofar")):
    if v in res: print(f"{name:54s} " + "  ".join(f"{b.get(k, 0) - res[v].get(k, 0):8.0f}" for k in keys))
PY
