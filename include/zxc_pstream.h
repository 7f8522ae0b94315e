/* zxc_pstream.h — push streaming: the caller feeds input chunks and drains output chunks, nothing blocks on a FILE*.
 * Same names, signatures, return conventions and state-machine behaviour as the reference (include/zxc_pstream.h:82-292,
 * impl src/lib/zxc_pstream.c). The reference composes one zxc_compress_block / one block decode per full block on the calling
 * CPU thread; here a call hands EVERY complete block its input holds to the device (one wavefront per block, pieces of blocks
 * pipelined over the library's staging arenas), so the chunk a caller feeds per call is the batch the GPU works on: feed
 * zxc_cstream_in_size() / zxc_dstream_in_size() bytes (128 MiB, not one block) or more per call for throughput. A block is never
 * held back across calls: when a call returns 0, every block completed by its input has been compressed / decoded and drained,
 * like in the reference. Archives are byte for byte what zxc_compress() writes for the same options (non-seekable), whatever
 * the chunking. No CPU codec: without a HIP device the first call that has a block to process returns
 * ZXC_ERROR_GPU_UNAVAILABLE (sticky). One context, one thread at a time; the work runs on the calling thread's current device. */
#ifndef ZXC_PSTREAM_H
#define ZXC_PSTREAM_H
#include <stddef.h>
#include <stdint.h>
#include "zxc_export.h"
#include "zxc_opts.h"
#ifdef __cplusplus
extern "C" {
#endif

/* reference include/zxc_pstream.h:82-86 — the library advances pos as it consumes src[pos..size) */
typedef struct {
    const void* src;
    size_t size;
    size_t pos;
} zxc_inbuf_t;

/* reference include/zxc_pstream.h:100-104 — the library writes at dst + pos and advances pos; [dst + pos, dst + size) is
 * scratch during a call (decoded batches are copied from the device straight into it when they fit) */
typedef struct {
    void* dst;
    size_t size;
    size_t pos;
} zxc_outbuf_t;

typedef struct zxc_cstream_s zxc_cstream;
typedef struct zxc_dstream_s zxc_dstream;

/* reference :131 — level, block_size, checksum_enabled honoured (level clamped, a bad block_size fails); dictionary options
 * are refused (the push format carries no dict_id); NULL opts = defaults. NULL on failure. */
ZXC_EXPORT zxc_cstream* zxc_cstream_create(const zxc_compress_opts_t* opts);
/* reference :140 — NULL is a no-op */
ZXC_EXPORT void zxc_cstream_free(zxc_cstream* cs);
/* reference :171 — 0: in consumed and nothing pending; > 0: compressed bytes still pending (drain out, call again);
 * < 0: zxc_error_t, sticky */
ZXC_EXPORT int64_t zxc_cstream_compress(zxc_cstream* cs, zxc_outbuf_t* out, zxc_inbuf_t* in);
/* reference :190 — residual block, EOF block, footer; 0 = done (any later call: ZXC_ERROR_NULL_INPUT), > 0 pending */
ZXC_EXPORT int64_t zxc_cstream_end(zxc_cstream* cs, zxc_outbuf_t* out);
/* reference :200 / :211 — suggested chunk sizes (here: one batch window of source / its compressed bound); 0 for NULL */
ZXC_EXPORT size_t zxc_cstream_in_size(const zxc_cstream* cs);
ZXC_EXPORT size_t zxc_cstream_out_size(const zxc_cstream* cs);

/* reference :228 — only checksum_enabled is honoured; dictionary options are refused */
ZXC_EXPORT zxc_dstream* zxc_dstream_create(const zxc_decompress_opts_t* opts);
/* reference :235 */
ZXC_EXPORT void zxc_dstream_free(zxc_dstream* ds);
/* reference :261 — > 0: decoded bytes written by this call; 0: DONE, or no progress possible (more input needed);
 * < 0: zxc_error_t, sticky. Parses file header, blocks, EOF block, optional SEK block, footer; bytes behind a validated
 * footer are left in `in`. */
ZXC_EXPORT int64_t zxc_dstream_decompress(zxc_dstream* ds, zxc_outbuf_t* out, zxc_inbuf_t* in);
/* reference :274 — 1 once the footer has been validated */
ZXC_EXPORT int zxc_dstream_finished(const zxc_dstream* ds);
/* reference :282 / :292 — suggested chunk sizes (one batch window); 0 for NULL */
ZXC_EXPORT size_t zxc_dstream_in_size(const zxc_dstream* ds);
ZXC_EXPORT size_t zxc_dstream_out_size(const zxc_dstream* ds);

#ifdef __cplusplus
}
#endif
#endif
