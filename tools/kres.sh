#!/bin/bash
# tools/kres.sh [extra -D flags] : registers, spills, private segment and LDS of every decode kernel (device assembly only, CPU)
cd "$(dirname "$0")/../zxc_amd/csrc"
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -DZXC_EXPERIMENT "$@" -S --cuda-device-only -o /tmp/asm/dk.s zxc_decode_kernel.hip 2>/dev/null
grep -E "^\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):" /tmp/asm/dk.s | paste - - - - - - - | sed 's/\s\+/ /g; s/_segment_fixed_size//g; s/_count//g'
