// emu_decode_lean.cpp — the LEAN build of the decode kernel (zxc_decode_kernel.hip with -DZXC_LEAN_KERNEL: no PivCo section
// decoder) for the CPU wave emulator. Test infrastructure; driven by emu_decode_blocks_two_pass (emu_decode.cpp).
#include <hip/hip_runtime.h>
#include <stdint.h>
#define ZXC_LEAN_KERNEL 1
namespace lean_tu {  // (the kernel source defines non-inline device functions: keep this copy apart from emu_decode.cpp's)
#include "zxc_decode_kernel.hip"
}
