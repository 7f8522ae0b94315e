/* zxc_dev.h — types shared by the HIP kernels and the host-side C-ABI shim. */
#ifndef ZXC_DEV_H
#define ZXC_DEV_H
#include "../../include/zxc_mi355x.h" /* zxc_dev_job_t */

/* Device-side status that is not a reference zxc_error_t value: the block is valid
 * but uses a section coding this kernel cannot decode yet (maps to
 * ZXC_ERROR_GPU_UNSUPPORTED on the host; never a silent CPU fallback). */
#define ZXC_DEV_E_UNSUPPORTED (-101)
#define ZXC_DEV_DEFER (-103)      /* lean kernel only, never stored: the block goes to the full kernel's list */
#define ZXC_DEV_E_INTERNAL (-102) /* kernel self-check tripped (a bug, never an input property) */

/* Levels 6-7, launches without a dictionary: the launch-order pass sorts every block into one of three classes. PRE blocks
 * have their PivCo sections decoded by zxc_pivco_sections_kernel into a per-launch scratch buffer (offsets in 16-byte units)
 * and are then executed by the lean kernel like blocks with raw sections; pre[b].rc is the sections' verdict. */
#define ZXC_DEV_CLS_LEAN 0u /* raw sections (and every block the lean kernel can name an error for) */
#define ZXC_DEV_CLS_FULL 1u /* RLE literals, oversized or malformed coded sections: the one-wave full kernel */
#define ZXC_DEV_CLS_PRE 2u
typedef struct {
    uint32_t lit_off; /* decoded literals at scratch + 16 * lit_off + 16 */
    uint32_t tok_off; /* decoded tokens at scratch + 16 * tok_off */
    int32_t rc;
    uint32_t cls;
} zxc_dev_pre_t;

/* Scratch slot of a block with a coded (RLE / PivCo) section, shared by the kernels, the shim's pool and the CPU emulator:
 * [0, R) expanded literals (from +16) | [R, 2R) the section decoder's odd-depth level buffer | [2R, stride) decoded tokens
 * (level 7); R = block_size + 64 bytes of slack. */
#define ZXC_DEV_SLOT_REGION(bs) ((bs) + 64u)
#define ZXC_DEV_SLOT_STRIDE(bs) ((2u * ZXC_DEV_SLOT_REGION(bs) + (bs) / 5u + 16u + 64u + 255u) & ~255u)
#endif
