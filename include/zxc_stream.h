/* zxc_stream.h — FILE*-flavoured entry points, same names and signatures as the reference
 * (include/zxc_stream.h:69-117). The reference runs a reader / worker-pool / writer ring
 * (src/lib/zxc_driver.c:627-1030, one block per job); here the "workers" are one GPU launch per
 * batch of blocks: the host reads a batch, the device decodes or encodes every block of it at once,
 * the host writes the results in order. No CPU codec: without a HIP device the calls return
 * ZXC_ERROR_GPU_UNAVAILABLE. */
#ifndef ZXC_STREAM_H
#define ZXC_STREAM_H
#include <stdint.h>
#include <stdio.h>
#include "zxc_export.h"
#include "zxc_opts.h"
#include "zxc_seekable.h"
#ifdef __cplusplus
extern "C" {
#endif

/* reference include/zxc_stream.h:69 — total compressed bytes written (f_out may be NULL: dry run, the size only), or a negative
 * zxc_error_t */
ZXC_EXPORT int64_t zxc_stream_compress(FILE* f_in, FILE* f_out, const zxc_compress_opts_t* opts);

/* reference include/zxc_stream.h:83 — total decompressed bytes written (f_out may be NULL: integrity
 * check only), or a negative zxc_error_t */
ZXC_EXPORT int64_t zxc_stream_decompress(FILE* f_in, FILE* f_out, const zxc_decompress_opts_t* opts);

/* reference include/zxc_stream.h:96 — size from the footer, file position restored */
ZXC_EXPORT int64_t zxc_stream_get_decompressed_size(FILE* f_in);

/* reference include/zxc_stream.h:117 — pread-backed reader handed to zxc_seekable_open_reader;
 * f must stay open for the handle's lifetime */
ZXC_EXPORT zxc_seekable* zxc_seekable_open_file(FILE* f);

#ifdef __cplusplus
}
#endif
#endif
