#!/bin/bash
# tools/ringprof.sh : the ring-size question of VERDICT r3 weak #2b on the GPU box. Times the lean-kernel variants (4 / 8 KiB ring at 4 waves
# per SIMD, each also with every match source taken from the ring = no read-back at all), then counts instructions and L1 -> L2 requests
# of the two ring sizes (one --pmc pass each). Needs zxc_amd/libzxc_{allnear,r4kw4,r8kw4,r4kw4an,r8kw4an}.so (tools/build_variant.sh).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export AB_TILES=${AB_TILES:-10} AB_TIMEOUT=100
timeout 400 python $R/tools/abbench.py libzxc_mi355x.so libzxc_allnear.so libzxc_r4kw4.so libzxc_r8kw4.so libzxc_r4kw4an.so libzxc_r8kw4an.so libzxc_mi355x.so 2>&1 | grep "GB/s\|TIMEOUT" > $R/gpurun_out/r4e_ring_times.log
for v in r4kw4 r8kw4; do
  ZXC_TOOLS_AB=1 ZXC_LIB_VARIANT=libzxc_$v.so timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $R/gpurun_out/r4e_sq_$v -o s --output-format csv -- python $R/tools/abbench.py --one > $R/gpurun_out/r4e_sq_$v.log 2>&1
  ZXC_TOOLS_AB=1 ZXC_LIB_VARIANT=libzxc_$v.so timeout 150 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $R/gpurun_out/r4e_mem_$v -o m --output-format csv -- python $R/tools/abbench.py --one > $R/gpurun_out/r4e_mem_$v.log 2>&1
done
cat $R/gpurun_out/r4e_ring_times.log
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for v in ("r4kw4", "r8kw4"):
    for kind in ("sq", "mem"):
        tot = collections.defaultdict(float); n = collections.Counter()
        for f in glob.glob(f"{R}/gpurun_out/r4e_{kind}_{v}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Kernel_Name", "").startswith("zxc_decode_blocks_lean_kernel"):
                    tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
        for k in sorted(tot): print(f"{v} {k:28s} per launch {tot[k] / max(n[k], 1):16.0f}  per block {tot[k] / max(n[k], 1) / 32340:10.1f}  ({n[k]} launches)")
PY
