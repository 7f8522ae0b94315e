/* zxc_seekable.h — Seekable (random-access) API of libzxc_mi355x.so: the entry point
 * of the headline benchmark. Same contract as reference include/zxc_seekable.h; the
 * range decode batches every covered block into ONE kernel launch (one wavefront per
 * block) instead of the reference's per-thread block stripes
 * (src/lib/zxc_seekable.c:902-981). */
#ifndef ZXC_SEEKABLE_H
#define ZXC_SEEKABLE_H
#include <stddef.h>
#include <stdint.h>
#include "zxc_export.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct zxc_seekable_s zxc_seekable;

/* reference include/zxc_seekable.h:88 — borrows src for the handle's lifetime; NULL on failure */
ZXC_EXPORT zxc_seekable* zxc_seekable_open(const void* src, const size_t src_size);

/* reference include/zxc_seekable.h:120-137 */
typedef struct {
    int64_t (*read_at)(void* ctx, void* dst, size_t len, uint64_t offset);
    void* ctx;
    uint64_t size;
} zxc_reader_t;
ZXC_EXPORT zxc_seekable* zxc_seekable_open_reader(const zxc_reader_t* r);

/* reference include/zxc_seekable.h:145-175 */
ZXC_EXPORT uint32_t zxc_seekable_get_num_blocks(const zxc_seekable* s);
ZXC_EXPORT uint64_t zxc_seekable_get_decompressed_size(const zxc_seekable* s);
ZXC_EXPORT uint32_t zxc_seekable_get_block_comp_size(const zxc_seekable* s, const uint32_t block_idx);
ZXC_EXPORT uint32_t zxc_seekable_get_block_decomp_size(const zxc_seekable* s, const uint32_t block_idx);

/* reference include/zxc_seekable.h:191 / :214 — returns len or a negative zxc_error_t.
 * _mt: a block is a wavefront's work here, so host threads add nothing on ONE device and by default the call stays on the calling
 * thread's current device whatever n_threads says (a rank of a one-process-per-GPU job never touches the other ranks' GPUs).
 * With ZXC_MI355X_DEVICES=<ordinals, comma-separated; one may repeat> the covered blocks are cut into min(n_threads, listed
 * devices) contiguous parts (n_threads == 0: all listed), one host thread + stream + staging arena per part (zxc_host.c). */
ZXC_EXPORT int64_t zxc_seekable_decompress_range(zxc_seekable* s, void* dst, const size_t dst_capacity,
                                                 const uint64_t offset, const size_t len);
ZXC_EXPORT int64_t zxc_seekable_decompress_range_mt(zxc_seekable* s, void* dst, const size_t dst_capacity,
                                                    const uint64_t offset, const size_t len, int n_threads);

/* reference include/zxc_seekable.h:226 */
ZXC_EXPORT void zxc_seekable_free(zxc_seekable* s);

/* reference include/zxc_seekable.h:243 (impl src/lib/zxc_seekable.c:1144-1174): dictionary for a
 * dict-compressed archive; content (and optional 128-byte shared table) are copied. */
ZXC_EXPORT int zxc_seekable_set_dict(zxc_seekable* s, const void* dict, size_t dict_size, const void* dict_huf);

/* reference include/zxc_seekable.h:263-272 (impl src/lib/zxc_seekable.c:172-214) */
ZXC_EXPORT int64_t zxc_write_seek_table(uint8_t* dst, const size_t dst_capacity,
                                        const uint32_t* comp_sizes, const uint32_t num_blocks);
ZXC_EXPORT size_t zxc_seek_table_size(const uint32_t num_blocks);

#ifdef __cplusplus
}
#endif
#endif
