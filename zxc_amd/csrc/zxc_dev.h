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

/* Levels 6-7, launches without a dictionary: the launch-order pass (zxc_order_scatter_kernel) sorts every block into one
 * of three classes. The PivCo sections of PRE blocks are decoded by the workgroup section kernels (zxc_pivco_dir.inc) into a
 * per-launch scratch buffer, one work record per section; the blocks are then executed by the lean kernel's second entry
 * like blocks with raw sections. rc_lit / rc_tok are the sections' verdicts (literals first, as the one-wave path orders them). */
#define ZXC_DEV_CLS_LEAN 0u /* raw sections (and every block the lean kernel can name an error for) */
#define ZXC_DEV_CLS_FULL 1u /* RLE literals, oversized or malformed coded sections: the one-wave full kernel */
#define ZXC_DEV_CLS_PRE 2u
#define ZXC_DEV_CLS_LEAN_RLE 3u /* RLE-coded literals, raw tokens, header valid: zxc_rle_expand_kernel expands the literals into the
                                 * block's share of the launch's RLE scratch (lit_off, 16-byte units; verdict in rc_lit), the lean
                                 * kernel then runs the block like one with raw sections */
typedef struct {
    uint32_t lit_off; /* decoded literals at scratch + 16 * lit_off + 16 */
    uint32_t tok_off; /* decoded tokens at scratch + 16 * tok_off */
    int16_t rc_lit, rc_tok;
    uint32_t cls;
} zxc_dev_pre_t;
typedef struct {          /* one coded section on a size class's work list */
    uint64_t src_off;     /* its payload in the compressed buffer */
    uint32_t psize, n;    /* payload bytes (128-byte header included), symbols */
    uint32_t out_off4;    /* decoded bytes at scratch + 4 * out_off4 */
    uint32_t rc_slot;     /* verdict: ((int16_t*)pre)[rc_slot] */
    uint32_t pad[2];
} zxc_dev_sec_t;
/* Control words behind the launch-order buffer's order[] (zxc_hip_shim.hip): */
/* trailer_bytes argument of the decode kernels: 4 = a per-block checksum trails every block; with this bit on top, the checksums are
 * verified by zxc_block_checksum_kernel beside the decode (nine blocks per wavefront) and zxc_checksum_merge_kernel writes the verdicts:
 * the decode kernels only account for the trailer's bytes (round 6) */
#define ZXC_DEV_TRAILER_ELSEWHERE 0x80000000u
#define ZXC_DEV_CTL_WORDS 32u
#define ZXC_DEV_CTL_PRE 0u     /* [0] PRE blocks listed, [1] next to hand out (lean kernel, second entry) */
#define ZXC_DEV_CTL_CURSOR 2u  /* scratch handed out so far, 16-byte units */
#define ZXC_DEV_CTL_WANTED 3u  /* blocks that qualify for PRE, scratch or no scratch (the next launch's plan: zxc_hip_shim.hip) */
#define ZXC_DEV_CTL_SEC 4u     /* [4 + 2 c] sections listed in size class c, [5 + 2 c] next to hand out */
#define ZXC_DEV_CTL_RLE_CURSOR 10u /* RLE scratch handed out so far, 16-byte units (saturating like CURSOR) */
#define ZXC_DEV_CTL_RLE_WANTED 11u /* RLE scratch all LEAN_RLE candidates of the launch would take (sizes the next launch's buffer) */
#define ZXC_DEV_CTL_RLE_LIST 12u   /* [12] LEAN_RLE blocks listed (job indices from the END of the PRE entries array, backwards), [13] next to hand out */

/* Scratch slot of a block with a coded (RLE / PivCo) section, shared by the kernels, the shim's pool and the CPU emulator:
 * [0, R) expanded literals (from +16) | [R, 2R) the section decoder's odd-depth level buffer | [2R, stride) decoded tokens
 * (level 7); R = block_size + 64 bytes of slack. */
#define ZXC_DEV_SLOT_REGION(bs) ((bs) + 64u)
#define ZXC_DEV_SLOT_STRIDE(bs) ((2u * ZXC_DEV_SLOT_REGION(bs) + (bs) / 5u + 16u + 64u + 255u) & ~255u)
#endif
