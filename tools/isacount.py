#!/usr/bin/env python3
"""Static instruction mix of a kernel's assembly between `; PHMARK n` markers (design tool, CPU only).
usage: isacount.py file.s kernel_name [first_line last_line]
Classes follow the issue costs measured by tools/src/issue_test.hip on gfx950: VOP2-style 32-bit-encoded integer ops issue in 2
clocks per wave64, everything else on the vector ALU (VOP3, DPP/SDWA, compares, 64-bit ops) in 4; SALU 1 per clock per CU."""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))  # (a kernel has several s_endpgm: early returns)
CHEAP = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32",
         "v_mov_b32", "v_min_u32", "v_max_u32", "v_min_i32", "v_max_i32", "v_not_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32",
         "v_subb_co_u32", "v_cndmask_b32", "v_bfrev_b32", "v_ffbh_u32", "v_ffbl_b32", "v_xnor_b32", "v_mul_u32_u24", "v_mul_i32_i24"}
def classify(op, text):
    if op.startswith("v_"):
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        if op.endswith("_e64") or op.endswith("_dpp") or op.endswith("_sdwa") or " dpp" in text or "row_" in text or "sdwa" in text: return "valu4"
        if base in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"): return "vlane"
        if base.startswith("v_cmp"): return "valu4"
        return "valu2" if base in CHEAP else "valu4"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "swait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "sbranch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_bpermute") or op.startswith("ds_permute"): return "bperm"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return "other"
seg = "pre"; counts = collections.OrderedDict(); ops = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r";\s*PHMARK (\d+)", t)
    if m: seg = "after PH" + m.group(1); continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
    op = t.split()[0]
    c = classify(op, t)
    counts.setdefault(seg, collections.Counter())[c] += 1
    ops[seg][op] += 1
tot = collections.Counter()
print(f"{'segment':14s} valu2 valu4 vlane  salu sbrch swait   lds bperm  vmem")
for s, c in counts.items():
    tot.update(c)
    print(f"{s:14s} " + " ".join(f"{c[k]:5d}" for k in ("valu2", "valu4", "vlane", "salu", "sbranch", "swait", "lds", "bperm", "vmem")))
print(f"{'total':14s} " + " ".join(f"{tot[k]:5d}" for k in ("valu2", "valu4", "vlane", "salu", "sbranch", "swait", "lds", "bperm", "vmem")))
if len(sys.argv) > 3:
    for s in sys.argv[3:]:
        print(s, ops["after PH" + s].most_common(40))
