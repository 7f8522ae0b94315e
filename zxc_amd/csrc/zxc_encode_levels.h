/* zxc_encode_levels.h — level -> encode kernel entry and search effort (shared by the shim and the test emulator).
 * Reference parameters: zxc_get_lz77_params, src/lib/zxc_internal.h:965-979; mapping: table in zxc_encode_kernel.hip. */
#ifndef ZXC_ENCODE_LEVELS_H
#define ZXC_ENCODE_LEVELS_H
#include <stdint.h>
/* entry: 0 l1, 1 l2, 2 l3, 3 l4, 4 l57 (level 5), 5 l67 (levels 6-7: the 2^15-entry chain ring) (zxc_encode_blocks_kernel_*) */
typedef struct { int entry; uint32_t depth, sufficient, lazy, huf; } zxc_enc_level_t; /* huf: 0 none, 1 PivCo literals, 2 + tokens */
#define ZXC_ENC_PARSE_OPTIMAL 3u /* `lazy` value: no lazy probes, the price-based optimal parse (zxc_optparse.inc) picks the sequences */
static inline zxc_enc_level_t zxc_enc_level(int level) {
    static const zxc_enc_level_t t[8] = {
        {0, 1, 16, 0, 0},    /* (fallback = level 1) */
        {0, 1, 16, 0, 0},    /* 1: head only, GHI */
        {1, 3, 18, 0, 0},    /* 2: short chain, GHI */
        {2, 4, 16, 2, 0},    /* 3: the 20 KiB entry, four candidates in one round */
        {3, 6, 18, 2, 0},    /* 4 */
        {4, 18, 256, 2, 0},  /* 5 */
        {5, 33, 256, ZXC_ENC_PARSE_OPTIMAL, 1},  /* 6: optimal parse + PivCo-coded literal section */
        {5, 66, 256, 2, 2},  /* 7: + PivCo-coded token section. Lazy parse: with 66 candidates per position the optimal parse measured
                              *    the same sizes (+-0.15 %) for 0.64 x the speed (profiles/r4g_optimal_parse.log) */
    };
    return t[level < 1 ? 1 : (level > 7 ? 7 : level)];
}
/* The entry also depends on the block size (round 6): the 20 / 24 KiB tables of levels 3-4 carry a chain ring of 2^11 / 2^12 positions,
 * sized for 64 KiB blocks. In the reference's default 512 KiB blocks (and up to 2 MiB) a position's chain reaches 65 536 bytes back
 * (ZXC_LZ_WINDOW_SIZE, src/lib/zxc_common.c:197-199): there levels 3-5 take the 80 KiB entry of levels 6-7 (ring 2^15) at their own search effort: on text 1.005 x the reference's size at 512 KiB blocks (2^11: 1.093 x, 2^12: 1.078 x, 2^14: 1.030 x). */
static inline zxc_enc_level_t zxc_enc_level_bs(int level, uint32_t block_size) {
    zxc_enc_level_t p = zxc_enc_level(level);
    if (p.entry >= 2 && p.entry <= 4 && block_size > 65536u) p.entry = 5;
    return p;
}
#endif
