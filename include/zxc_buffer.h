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
 * compress into a v8 archive (optionally seekable). Blocks are encoded on the GPU by one
 * match-finding strategy whatever the level (valid, reference-decodable output; sizes differ
 * from the CPU encoder's). Returns archive size or a negative zxc_error_t. */
ZXC_EXPORT int64_t zxc_compress(const void* src, const size_t src_size, void* dst,
                                const size_t dst_capacity, const zxc_compress_opts_t* opts);

/* reference include/zxc_buffer.h:140 (impl src/lib/zxc_dispatch.c:842-1005).
 * Whole-frame decode: returns decoded size or a negative zxc_error_t. */
ZXC_EXPORT int64_t zxc_decompress(const void* src, const size_t src_size, void* dst,
                                  const size_t dst_capacity, const zxc_decompress_opts_t* opts);

/* reference include/zxc_buffer.h:195 (impl src/lib/zxc_dispatch.c:1203-1225) */
ZXC_EXPORT uint64_t zxc_get_decompressed_size(const void* src, const size_t src_size);

#ifdef __cplusplus
}
#endif
#endif
