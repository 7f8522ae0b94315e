/* zxc_mi355x.h — device-resident C-ABI of libzxc_mi355x.so (plain pointers and
 * sizes; no C++/torch types). This is the boundary the reference's own block loops
 * would bind to run on an MI355X:
 *
 *   zxc_mi355x_decode_blocks_device  replaces the per-block calls to the internal
 *       zxc_decompress_chunk_wrapper (src/lib/zxc_dispatch.c:279-283, :479-493) made
 *       by zxc_decompress_frame (:912-1001), the seekable ST loop
 *       (src/lib/zxc_seekable.c:742-781) and the seekable MT worker (:943-976):
 *       instead of one 64 KiB block per call on a CPU thread, one launch decodes a
 *       whole table of independent blocks, one wavefront per block.
 *   zxc_mi355x_plan_seekable  produces that table from a seekable handle
 *       (the job planning of src/lib/zxc_seekable.c:1033-1056).
 *
 * All d_* arguments are device pointers on the current HIP device. Return values
 * follow the zxc convention: >= 0 success, < 0 zxc_error_t (zxc_error.h). */
#ifndef ZXC_MI355X_H
#define ZXC_MI355X_H
#include <stddef.h>
#include <stdint.h>
#include "zxc_export.h"
#include "zxc_seekable.h"
#ifdef __cplusplus
extern "C" {
#endif

/* One block to decode (24 bytes, device- and host-visible layout). */
typedef struct zxc_dev_job {
    uint64_t comp_off;  /* byte offset of the block's 8-byte header inside d_comp */
    uint64_t out_off;   /* byte offset inside d_out; MUST be a multiple of 16 */
    uint32_t comp_size; /* physical block size: header + payload (+ 4-byte checksum trailer) */
    uint32_t out_len;   /* decoded bytes to keep: block_size, or the archive's tail remainder */
} zxc_dev_job_t;

/* Number of usable HIP devices (0 when there is none / no driver). */
ZXC_EXPORT int zxc_mi355x_device_count(void);
/* Select the device used by the calling thread (hipSetDevice). */
ZXC_EXPORT int zxc_mi355x_set_device(int device);
/* The device the calling thread uses (hipGetDevice), or a negative zxc_error_t. */
ZXC_EXPORT int zxc_mi355x_get_device(void);

/* Device memory helpers so a C caller needs no HIP headers. */
ZXC_EXPORT void* zxc_mi355x_malloc(size_t bytes);
ZXC_EXPORT void zxc_mi355x_free(void* d_ptr);
ZXC_EXPORT int zxc_mi355x_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
ZXC_EXPORT int zxc_mi355x_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
ZXC_EXPORT int zxc_mi355x_synchronize(void* stream);
/* Gives back the device memory the library keeps between calls (staging arenas of the host API that nobody is using;
 * on the calling thread's device also the section decoders' scratch pools and the launch-order buffers). Call it when
 * no launch of this library is in flight on that device. Arenas are bounded anyway: every host entry point works in
 * batches of 256 MiB of output, and a buffer that grew past 768 MiB is freed when its call ends. */
ZXC_EXPORT void zxc_mi355x_release_cached(void);

/* Fill jobs[0..n_blocks) for blocks [first_block, first_block + n_blocks) of an open
 * seekable archive. Block i's compressed bytes are expected at
 * d_comp + (archive_offset(i) - comp_rebase) and its output at
 * d_out + (i - first_block) * block_size (16-byte aligned by construction).
 * Pass comp_rebase = 0 when the whole archive was uploaded. Returns n_blocks. */
ZXC_EXPORT int64_t zxc_mi355x_plan_seekable(const zxc_seekable* s, uint32_t first_block,
                                            uint32_t n_blocks, uint64_t comp_rebase,
                                            zxc_dev_job_t* jobs);

/* Decode n_jobs independent blocks, asynchronously on `stream` (a hipStream_t, or
 * NULL for the default stream). d_status[i] receives block i's decoded size or a
 * negative zxc_error_t. Requirements: d_out + job.out_off 16-byte aligned; d_out
 * readable/writable up to round_up(out_off + out_len, 16) + 16. block_size is the
 * archive's block size (bounds scratch and the per-block output cap
 * block_size + 2112, like the reference). verify_trailer = 1 means every block
 * carries its 4-byte checksum trailer and it is verified on device (rapidhash of the
 * payload, ZXC_ERROR_BAD_CHECKSUM on mismatch). */
ZXC_EXPORT int zxc_mi355x_decode_blocks_device(const void* d_comp, const zxc_dev_job_t* d_jobs,
                                               uint32_t n_jobs, void* d_out, int32_t* d_status,
                                               uint32_t block_size, int verify_trailer, void* stream);

/* Same, for archives compressed with a dictionary: d_dict[0..dict_size) is logically prepended to
 * every block (reference d_floor = dst - dict_size, src/lib/zxc_decompress.c:1028); d_dict_huf is
 * the dictionary's 128-byte shared literal table (enc_lit = 3 sections) or NULL. */
ZXC_EXPORT int zxc_mi355x_decode_blocks_dict_device(const void* d_comp, const zxc_dev_job_t* d_jobs,
                                                    uint32_t n_jobs, void* d_out, int32_t* d_status,
                                                    uint32_t block_size, int verify_trailer,
                                                    const void* d_dict, uint32_t dict_size,
                                                    const void* d_dict_huf, void* stream);

/* ---- encode side (LZ77 match finder + GLO serialiser, zxc_amd/csrc/zxc_encode_kernel.hip) ----
 * Replaces the per-block calls to zxc_compress_chunk_wrapper (src/lib/zxc_compress.c:2041-2074)
 * made by zxc_compress (src/lib/zxc_dispatch.c:734-780). Block i of the source
 * (d_src + i*block_size) becomes one complete v8 block (8-byte header + payload, GLO or RAW) at
 * d_slots + i*zxc_mi355x_encode_slot_stride(block_size); its size lands in d_sizes[i]
 * (= the seek-table entry; with_checksum appends the 4-byte rapidhash trailer). Asynchronous on `stream`.
 * d_src must be READABLE up to src_size + 32 (the match finder compares 16 bytes at a time and clamps lengths to the
 * block afterwards; the bytes themselves are never used). */
ZXC_EXPORT uint32_t zxc_mi355x_encode_slot_stride(uint32_t block_size);
ZXC_EXPORT int zxc_mi355x_encode_blocks_device(const void* d_src, uint64_t src_size, uint32_t block_size,
                                               int level, int with_checksum, void* d_slots,
                                               uint32_t* d_sizes, void* stream);
/* Same with a dictionary (reference: opts.dict of zxc_compress, src/lib/zxc_dispatch.c:700-733, and the [dict | block]
 * buffer of zxc_compress_block :1688-1697): every block's tables are seeded with d_dict[0..dict_size), matches may reach
 * into it. d_work is scratch of zxc_mi355x_encode_dict_work_size() bytes (one [dict | block] image per block). */
ZXC_EXPORT uint64_t zxc_mi355x_encode_dict_work_size(uint64_t src_size, uint32_t block_size, uint32_t dict_size);
ZXC_EXPORT int zxc_mi355x_encode_blocks_dict_device(const void* d_src, uint64_t src_size, uint32_t block_size,
                                                    int level, int with_checksum, const void* d_dict,
                                                    uint32_t dict_size, void* d_work, void* d_slots,
                                                    uint32_t* d_sizes, void* stream);
/* Compaction: block i's d_sizes[i] bytes go to d_out + d_offsets[i] (prefix sums computed by the
 * caller, like seek_comp[] in src/lib/zxc_dispatch.c:761-776). */
ZXC_EXPORT int zxc_mi355x_gather_blocks_device(const void* d_slots, uint32_t block_size,
                                               const uint32_t* d_sizes, const uint64_t* d_offsets,
                                               void* d_out, uint32_t n_blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif
