/* zxc_opts.h — option structs, byte-identical layout to the reference
 * (include/zxc_opts.h:58-95; size guards :105-111) so FFI callers need no change. */
#ifndef ZXC_OPTS_H
#define ZXC_OPTS_H
#include <stddef.h>
#include <stdint.h>
#include "zxc_export.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef void (*zxc_progress_callback_t)(uint64_t bytes_processed, uint64_t bytes_total,
                                        const void* user_data);

typedef struct {
    int n_threads;        /* ignored: parallelism is the GPU's */
    int level;            /* 1..7, 0 = default (3); device encoder effort per level: INTEGRATION.md / zxc_encode_levels.h */
    size_t block_size;    /* power of two in [4 KiB, 2 MiB], 0 = 512 KiB */
    int checksum_enabled;
    int seekable;
    const void* dict;
    size_t dict_size;
    const void* dict_huf;
    zxc_progress_callback_t progress_cb;
    void* user_data;
} zxc_compress_opts_t;

typedef struct {
    int n_threads;        /* ignored */
    int checksum_enabled; /* verify per-block + global checksums */
    const void* dict;
    size_t dict_size;
    const void* dict_huf;
    zxc_progress_callback_t progress_cb;
    void* user_data;
} zxc_decompress_opts_t;

ZXC_EXPORT size_t zxc_compress_opts_size(void);
ZXC_EXPORT size_t zxc_decompress_opts_size(void);

#ifdef __cplusplus
}
#endif
#endif
