/* zxc_buffer.h — Buffer API of libzxc_mi355x.so. Same names, argument meaning and
 * error behaviour as the reference Buffer API (include/zxc_buffer.h); every
 * block is decoded by the HIP kernels (zxc_amd/csrc/zxc_decode_kernel.hip).
 * Caller owns src and dst; both are host pointers. */
#ifndef ZXC_BUFFER_H
#define ZXC_BUFFER_H
#include <stddef.h>
#include <stdint.h>
#include "zxc_export.h"
#include "zxc_opts.h"
#ifdef __cplusplus
extern "C" {
#endif

/* reference include/zxc_buffer.h:60-80 */
ZXC_EXPORT int zxc_min_level(void);
ZXC_EXPORT int zxc_max_level(void);
ZXC_EXPORT int zxc_default_level(void);
ZXC_EXPORT const char* zxc_version_string(void);

/* reference include/zxc_buffer.h:98 (impl src/lib/zxc_common.c:850-862) */
ZXC_EXPORT uint64_t zxc_compress_bound(const size_t input_size);

/* reference include/zxc_buffer.h:119 (impl src/lib/zxc_dispatch.c:658-818). Whole-buffer
 * compress into a v8 archive (optionally seekable). Blocks are encoded on the GPU by a per-level hash-chain match
 * finder (depth / sufficient length / lazy probes per level: zxc_amd/csrc/zxc_encode_levels.h; GHI blocks at levels
 * 1-2, GLO above, PivCo-coded sections at levels 6-7, opts->dict seeds every block's tables). The output is a valid
 * archive the unmodified reference decodes; its bytes differ from the CPU encoder's (another parse). Returns archive
 * size or a negative zxc_error_t. */
ZXC_EXPORT int64_t zxc_compress(const void* src, const size_t src_size, void* dst,
                                const size_t dst_capacity, const zxc_compress_opts_t* opts);

/* reference include/zxc_buffer.h:140 (impl src/lib/zxc_dispatch.c:842-1005).
 * Whole-frame decode: returns decoded size or a negative zxc_error_t. */
ZXC_EXPORT int64_t zxc_decompress(const void* src, const size_t src_size, void* dst,
                                  const size_t dst_capacity, const zxc_decompress_opts_t* opts);

/* reference include/zxc_buffer.h:155 / :181 (impl src/lib/zxc_dispatch.c:1129-1185): decode inside ONE caller buffer that
 * holds the archive flush-right; the bound = decoded size + one block + per-block overhead + trailer + 2112 (or
 * archive size + one block + 2112, whichever is larger), the reference's formula */
ZXC_EXPORT size_t zxc_decompress_inplace_bound(const void* src, const size_t src_size);
ZXC_EXPORT int64_t zxc_decompress_inplace(void* buffer, const size_t buffer_capacity, const size_t comp_size,
                                          const zxc_decompress_opts_t* opts);

/* reference include/zxc_buffer.h:195 (impl src/lib/zxc_dispatch.c:1203-1225) */
ZXC_EXPORT uint64_t zxc_get_decompressed_size(const void* src, const size_t src_size);

/* reference include/zxc_buffer.h:204 (impl src/lib/zxc_dispatch.c:1234-1242): dictionary id of an archive header, 0 = none */
ZXC_EXPORT uint32_t zxc_get_dict_id(const void* src, size_t src_size);

/* ---- Block API (no file framing): reference include/zxc_buffer.h:236-362, impl src/lib/zxc_dispatch.c:1627-1858,
 * bounds src/lib/zxc_common.c:873-902. One block per call = one single-workgroup launch (correct, not fast:
 * bulk callers use zxc_decompress / the seekable API / zxc_mi355x.h, which decode all blocks in one launch). */
typedef struct zxc_cctx_s zxc_cctx; /* reference include/zxc_buffer.h:236 */
typedef struct zxc_dctx_s zxc_dctx; /* :238 */
ZXC_EXPORT uint64_t zxc_compress_block_bound(size_t input_size);              /* :255  8 + n + 68 + 4, 0 if n = 0 or > 2 MiB */
ZXC_EXPORT uint64_t zxc_decompress_block_bound(const size_t uncompressed_size); /* :272  n + 2112, 0 if n > 2 MiB */
/* :382  estimated peak working memory of one zxc_compress_block call (here: device memory of the staging arena) */
ZXC_EXPORT uint64_t zxc_estimate_cctx_size(size_t src_size, int level);
/* :300  header(8) + payload [+ checksum(4)]; level, block_size, checksum_enabled and dict / dict_size of opts are used
 * (a dictionary seeds the block's match-finder tables, reference src/lib/zxc_dispatch.c:1688-1697). The context's sticky
 * block size is the caller's block_size, not the reference's dictionary-padded effective size (:1650): nothing outside
 * the context can observe it. */
ZXC_EXPORT int64_t zxc_compress_block(zxc_cctx* cctx, const void* src, size_t src_size, void* dst,
                                      size_t dst_capacity, const zxc_compress_opts_t* opts);
/* :328  dst_capacity in [decoded size, 2 MiB + 2112]; decodes with capacity block_size_ceil(dst_capacity) + 2112 */
ZXC_EXPORT int64_t zxc_decompress_block(zxc_dctx* dctx, const void* src, size_t src_size, void* dst,
                                        size_t dst_capacity, const zxc_decompress_opts_t* opts);
/* :362  strict variant: dst_capacity may equal the decoded size exactly (<= 2 MiB) */
ZXC_EXPORT int64_t zxc_decompress_block_safe(zxc_dctx* dctx, const void* src, const size_t src_size,
                                             void* dst, const size_t dst_capacity,
                                             const zxc_decompress_opts_t* opts);

/* ---- reusable contexts: reference include/zxc_buffer.h:414-486 (impl src/lib/zxc_dispatch.c:1262-1600). Here they
 * only carry the sticky options; the working memory is the library's per-device arena. */
ZXC_EXPORT zxc_cctx* zxc_create_cctx(const zxc_compress_opts_t* opts); /* :414  NULL on a bad block size */
ZXC_EXPORT void zxc_free_cctx(zxc_cctx* cctx);                         /* :421 */
ZXC_EXPORT int64_t zxc_compress_cctx(zxc_cctx* cctx, const void* src, size_t src_size, void* dst,
                                     size_t dst_capacity, const zxc_compress_opts_t* opts); /* :448 (sticky opts) */
ZXC_EXPORT zxc_dctx* zxc_create_dctx(void);                            /* :461 */
ZXC_EXPORT void zxc_free_dctx(zxc_dctx* dctx);                         /* :468 */
ZXC_EXPORT int64_t zxc_decompress_dctx(zxc_dctx* dctx, const void* src, size_t src_size, void* dst,
                                       size_t dst_capacity, const zxc_decompress_opts_t* opts); /* :486 */

/* ---- static contexts: reference include/zxc_buffer.h:494-604 (impl src/lib/zxc_dispatch.c:1860-1965). The context lives in a
 * buffer the caller owns; zxc_free_* are no-ops on it. The reference carves its tables out of that buffer; here they are in LDS /
 * device memory, the workspace holds the handle (one cache line, two when carved for levels 6-7). Same contract: block_size
 * locked (another one: ZXC_ERROR_BAD_BLOCK_SIZE), a raise into levels 6-7 on a workspace carved below them: ZXC_ERROR_BAD_LEVEL,
 * level / checksum otherwise per call. */
ZXC_EXPORT size_t zxc_static_cctx_workspace_size(const size_t block_size, const int level); /* :540  0 on invalid arguments */
ZXC_EXPORT zxc_cctx* zxc_init_static_cctx(void* workspace, const size_t workspace_size,
                                          const zxc_compress_opts_t* opts);                  /* :567  NULL: too small / invalid */
ZXC_EXPORT size_t zxc_static_dctx_workspace_size(const size_t block_size);                   /* :581 */
ZXC_EXPORT zxc_dctx* zxc_init_static_dctx(void* workspace, const size_t workspace_size,
                                          const size_t block_size);                          /* :601 */

#ifdef __cplusplus
}
#endif
#endif
