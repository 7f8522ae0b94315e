/* zxc_dict.h — dictionary identity and the .zxd container (content + 128-byte shared literal table), host-side
 * helpers of libzxc_mi355x.so with the reference's names and behaviour (reference include/zxc_dict.h:72-129, :204;
 * impl src/lib/zxc_dict.c:35-205). Dictionary TRAINING (zxc_train_dict / zxc_train_dict_huf / zxc_dict_train) is out
 * of this library's scope: train with the reference, load the .zxd here. */
#ifndef ZXC_DICT_H
#define ZXC_DICT_H
#include <stddef.h>
#include <stdint.h>
#include "zxc_export.h"
#ifdef __cplusplus
extern "C" {
#endif

/* reference include/zxc_dict.h:72 — id of (content [, table]): the value the archive header carries; 0 for no dictionary */
ZXC_EXPORT uint32_t zxc_dict_id(const void* dict, size_t dict_size, const void* huf_lengths);
/* :91 — parse a .zxd buffer into in-buffer views (checks magic, version, header CRC16 and the id of the pair) */
ZXC_EXPORT int zxc_dict_load(const void* buf, size_t buf_size, const void** content_out, size_t* content_size_out,
                             const void** huf_out, uint32_t* dict_id_out);
/* :108 / :117 — serialise content + table as a .zxd; bound = 16 + content + 128 */
ZXC_EXPORT int64_t zxc_dict_save(const void* content, size_t content_size, const void* huf_lengths, void* buf,
                                 size_t buf_capacity);
ZXC_EXPORT size_t zxc_dict_save_bound(size_t content_size);
/* :129 — id stored in a .zxd header (0 if not a .zxd) */
ZXC_EXPORT uint32_t zxc_dict_get_id(const void* buf, size_t buf_size);
/* :146 — dictionary content from samples: the byte sequences that cover most of the corpus' k-grams (host arithmetic; the
 * reference's bytes). Size of the content or a negative zxc_error_t */
ZXC_EXPORT int64_t zxc_train_dict(const void* const* samples, const size_t* sample_sizes, size_t n_samples, void* dict_buf,
                                  size_t dict_capacity);
/* :168 — the shared literal table for a trained dictionary: the samples' 4 KiB slices are compressed against it ON THE DEVICE (one
 * launch) and the literals the parser leaves are histogrammed; a valid 128-byte table for either library, not the reference's
 * bytes (another parser leaves other literals) */
ZXC_EXPORT int zxc_train_dict_huf(const void* const* samples, const size_t* sample_sizes, size_t n_samples, const void* dict,
                                  size_t dict_size, uint8_t* huf_lengths_out);
/* :191 — both, serialised as a .zxd */
ZXC_EXPORT int64_t zxc_dict_train(const void* const* samples, const size_t* sample_sizes, size_t n_samples, void* zxd_buf,
                                  size_t zxd_capacity);
/* :204 — the 128-byte table inside a .zxd buffer (NULL if not a .zxd) */
ZXC_EXPORT const void* zxc_dict_huf(const void* buf, size_t buf_size);

#ifdef __cplusplus
}
#endif
#endif
