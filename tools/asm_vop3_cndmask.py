#!/usr/bin/env python3
"""Assembly post-pass (stdin -> stdout): re-encode VOP2 `v_cndmask_b32_e32 vD, src0, vB, vcc` as VOP3 `v_cndmask_b32_e64`.
Measured on gfx950 (tools/src/vcc_test.hip, profiles/r3_vcc_test.log): the VOP2 form with its implicit VCC costs ~17-23 clocks per
wave64 instruction unless it directly follows the v_cmp that wrote VCC; the VOP3 form costs ~4 whatever produced the mask.
Only operands VOP3 can encode on gfx9 are converted (registers and inline constants; a 32-bit literal stays VOP2)."""
import re, sys
INLINE = re.compile(r"^(-?\d+|0x[0-9a-fA-F]+|-?\d+\.\d+|v\d+|s\d+|vcc_lo|vcc_hi|m0|exec_lo|exec_hi)$")
def inline_ok(tok):
    if re.match(r"^(v\d+|s\d+|vcc_lo|vcc_hi)$", tok): return True
    if re.match(r"^-?\d+$", tok): return -16 <= int(tok) <= 64
    if re.match(r"^0x[0-9a-fA-F]+$", tok): return int(tok, 16) <= 64
    return tok in ("0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0")
n = k = 0
for line in sys.stdin:
    m = re.match(r"^(\s*)v_cndmask_b32_e32 (v\d+), ([^,]+), (v\d+), vcc(\s*(;.*)?)$", line.rstrip("\n"))
    if m:
        n += 1
        if inline_ok(m.group(3).strip()):
            k += 1
            line = f"{m.group(1)}v_cndmask_b32_e64 {m.group(2)}, {m.group(3)}, {m.group(4)}, vcc{m.group(5)}\n"
    sys.stdout.write(line)
sys.stderr.write(f"asm_vop3_cndmask: {k} of {n} VOP2 v_cndmask re-encoded\n")
