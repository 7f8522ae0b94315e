/*
 * zxc_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the ZXC (format v8) *decode* path, used solely as the
 * checker for the HIP kernels: tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product library (zxc_amd/csrc) never does.
 *
 * Parity status: PINNED. The restatement is checked against the reference's
 * conformance vectors (conformance/valid + invalid with pinned error codes),
 * the frozen golden archives (tests/format/golden) and, differentially, against
 * the unmodified reference compiled by oracle/Makefile into oracle/_ref.
 *
 * Error codes are the reference's zxc_error_t values (include/zxc_error.h:38-74).
 */
#ifndef ZXC_ORACLE_H
#define ZXC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ZXO_OK = 0,
    ZXO_E_MEMORY = -1,
    ZXO_E_DST_TOO_SMALL = -2,
    ZXO_E_SRC_TOO_SMALL = -3,
    ZXO_E_BAD_MAGIC = -4,
    ZXO_E_BAD_VERSION = -5,
    ZXO_E_BAD_HEADER = -6,
    ZXO_E_BAD_CHECKSUM = -7,
    ZXO_E_CORRUPT_DATA = -8,
    ZXO_E_BAD_OFFSET = -9,
    ZXO_E_OVERFLOW = -10,
    ZXO_E_IO = -11,
    ZXO_E_NULL_INPUT = -12,
    ZXO_E_BAD_BLOCK_TYPE = -13,
    ZXO_E_BAD_BLOCK_SIZE = -14,
    ZXO_E_DICT_REQUIRED = -15,
    ZXO_E_DICT_MISMATCH = -16,
    ZXO_E_DICT_TOO_LARGE = -17,
};

/* Decode-side context: what the reference keeps in zxc_cctx_t for decoding
 * (src/lib/zxc_internal.h:1633-1690): block size (sizes the literal scratch),
 * checksum switch, optional dictionary prefix and shared literal table. */
typedef struct {
    uint32_t block_size;       /* from the file header; bounds n_literals / n_sequences scratch */
    int checksum_enabled;      /* verify the 4-byte trailer of each block */
    const uint8_t* dict;       /* dictionary content or NULL */
    size_t dict_size;
    const uint8_t* dict_huf;   /* 128-byte shared literal code lengths or NULL */
    int strict_tail;           /* 1: the reference's "safe" decoders (zxc_decompress_block_safe: exact capacity, a 4x batch
                                * that would overflow rolls back to the exact loops); 0: every other entry point, whose 4x
                                * batches answer OVERFLOW from their output reserve (zxc_decompress.c:626-656) */
} zxo_ctx_t;

typedef struct {
    uint32_t n_blocks;
    uint32_t block_size;
    uint64_t total_decomp;
    int has_checksum;
    uint32_t dict_id;
    uint32_t* comp_sizes;    /* [n_blocks]   physical block sizes (header + payload + trailer) */
    uint64_t* comp_offsets;  /* [n_blocks+1] byte offsets into the archive */
} zxo_seek_table_t;

/* per-block stream statistics, for design notes / bench roofline accounting */
typedef struct {
    uint32_t type, n_sequences, n_literals, enc_lit, enc_tok, enc_off;
    uint32_t lit_bytes, tok_bytes, off_bytes, extra_bytes, n_varints;
    uint64_t match_bytes, off_hist[17]; /* match bytes by ceil(log2(offset)) */
    uint32_t overlap_matches;           /* off < ml */
} zxo_block_stats_t;

uint8_t zxo_hash8(const uint8_t hdr8[8]);
uint16_t zxo_hash16(const uint8_t hdr16[16]);
uint32_t zxo_checksum32(const void* data, size_t len);          /* rapidhash v3 folded to 32 bits */
uint64_t zxo_rapidhash(const void* data, size_t len, uint64_t seed);

int zxo_read_file_header(const uint8_t* src, size_t n, uint32_t* block_size, int* has_checksum,
                         uint32_t* dict_id);

/* One physical block (8-byte header + payload [+ trailer]) -> bytes. Mirrors
 * zxc_decompress_chunk_wrapper (src/lib/zxc_decompress.c:1646-1695). Returns
 * decoded size or a negative error. dst_cap is the logical capacity (the
 * reference always passes block_size + 2112). */
int zxo_decode_block(const zxo_ctx_t* ctx, const uint8_t* src, size_t src_sz, uint8_t* dst,
                     size_t dst_cap);

/* Whole-frame decode, mirrors zxc_decompress (src/lib/zxc_dispatch.c:842-1005). */
int64_t zxo_decompress(const void* src, size_t src_size, void* dst, size_t dst_capacity,
                       int checksum_enabled, const void* dict, size_t dict_size,
                       const void* dict_huf);

/* Seek table, mirrors zxc_seekable_parse (src/lib/zxc_seekable.c:270-396). */
int zxo_seek_table_parse(const uint8_t* data, size_t n, zxo_seek_table_t* out);
void zxo_seek_table_free(zxo_seek_table_t* t);
int64_t zxo_seekable_decompress_range(const uint8_t* data, size_t n, const zxo_seek_table_t* t,
                                      void* dst, size_t dst_capacity, uint64_t offset, size_t len);

/* PivCo section (128-byte lengths header + node runs), src/lib/zxc_huffman.c:2447. */
int zxo_huf_decode_section(const uint8_t* payload, size_t payload_size, uint8_t* dst, size_t n);

int zxo_block_stats(const uint8_t* src, size_t src_sz, zxo_block_stats_t* st);

#ifdef __cplusplus
}
#endif
#endif
