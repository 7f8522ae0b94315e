/*
 * zxc_host.c — host side (plain C) of libzxc_mi355x.so: the reference's public
 * Buffer and Seekable APIs, re-implemented as *batched block loops* over the HIP
 * decode kernel. Container logic (headers, CRCs, seek table, footer checks, error
 * precedence) follows the reference; every block payload is decoded on the GPU —
 * there is no CPU decoder in this library and no fallback: without a usable HIP
 * device the calls fail with ZXC_ERROR_GPU_UNAVAILABLE.
 *
 *   zxc_compress                      <- src/lib/zxc_dispatch.c:658-818
 *   zxc_decompress                    <- src/lib/zxc_dispatch.c:842-1005
 *   zxc_get_decompressed_size         <- src/lib/zxc_dispatch.c:1203-1225
 *   zxc_seekable_open / _open_reader  <- src/lib/zxc_seekable.c:270-554
 *   zxc_seekable_decompress_range[_mt]<- src/lib/zxc_seekable.c:695-785, :999-1108
 *   zxc_write_seek_table / _size      <- src/lib/zxc_seekable.c:172-214
 */
#include <limits.h>
#include <pthread.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include "../../include/zxc.h"

#define MAGIC 0x9CB02EF5u
#define FORMAT_VERSION 8
#define BLK_HDR 8
#define TAIL_PAD 2112u /* ZXC_DECOMPRESS_TAIL_PAD, src/lib/zxc_internal.h:341 */
#define HOST_BATCH_BYTES ((size_t)256 << 20) /* output slots per launch of the host Buffer API */
#define FRAME_BATCH_BYTES ((size_t)128 << 20) /* zxc_decompress: two batches in flight (upload + launch of one beside the download of the other) */
enum { BLK_RAW = 0, BLK_GLO = 1, BLK_GHI = 2, BLK_SEK = 254, BLK_EOF = 255 };

static uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t rd32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
static void wr32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

/* header check bytes (src/lib/zxc_internal.h:1188-1214): xorshift of the LE words */
static uint64_t xs_mix(uint64_t h) {
    h ^= h << 13;
    h ^= h >> 7;
    h ^= h << 17;
    return h;
}
static uint8_t hdr_hash8(const uint8_t* p) {
    const uint64_t h = xs_mix(rd64(p) ^ 0x9E3779B97F4A7C15ull);
    return (uint8_t)((h >> 32) ^ h);
}
static uint16_t hdr_hash16(const uint8_t* p) {
    const uint64_t h = xs_mix(rd64(p) ^ rd64(p + 8) ^ 0xD2D84A61D2D84A61ull);
    const uint32_t r = (uint32_t)((h >> 32) ^ h);
    return (uint16_t)((r >> 16) ^ r);
}

/* ---------------------------------------------------------------- misc API */
const char* zxc_error_name(const int code) {
    switch (code) {
        case ZXC_OK: return "ZXC_OK";
        case ZXC_ERROR_MEMORY: return "ZXC_ERROR_MEMORY";
        case ZXC_ERROR_DST_TOO_SMALL: return "ZXC_ERROR_DST_TOO_SMALL";
        case ZXC_ERROR_SRC_TOO_SMALL: return "ZXC_ERROR_SRC_TOO_SMALL";
        case ZXC_ERROR_BAD_MAGIC: return "ZXC_ERROR_BAD_MAGIC";
        case ZXC_ERROR_BAD_VERSION: return "ZXC_ERROR_BAD_VERSION";
        case ZXC_ERROR_BAD_HEADER: return "ZXC_ERROR_BAD_HEADER";
        case ZXC_ERROR_BAD_CHECKSUM: return "ZXC_ERROR_BAD_CHECKSUM";
        case ZXC_ERROR_CORRUPT_DATA: return "ZXC_ERROR_CORRUPT_DATA";
        case ZXC_ERROR_BAD_OFFSET: return "ZXC_ERROR_BAD_OFFSET";
        case ZXC_ERROR_OVERFLOW: return "ZXC_ERROR_OVERFLOW";
        case ZXC_ERROR_IO: return "ZXC_ERROR_IO";
        case ZXC_ERROR_NULL_INPUT: return "ZXC_ERROR_NULL_INPUT";
        case ZXC_ERROR_BAD_BLOCK_TYPE: return "ZXC_ERROR_BAD_BLOCK_TYPE";
        case ZXC_ERROR_BAD_BLOCK_SIZE: return "ZXC_ERROR_BAD_BLOCK_SIZE";
        case ZXC_ERROR_DICT_REQUIRED: return "ZXC_ERROR_DICT_REQUIRED";
        case ZXC_ERROR_DICT_MISMATCH: return "ZXC_ERROR_DICT_MISMATCH";
        case ZXC_ERROR_DICT_TOO_LARGE: return "ZXC_ERROR_DICT_TOO_LARGE";
        case ZXC_ERROR_BAD_LEVEL: return "ZXC_ERROR_BAD_LEVEL";
        case ZXC_ERROR_GPU_UNAVAILABLE: return "ZXC_ERROR_GPU_UNAVAILABLE";
        case ZXC_ERROR_GPU_UNSUPPORTED: return "ZXC_ERROR_GPU_UNSUPPORTED";
        default: return "ZXC_UNKNOWN_ERROR";
    }
}
int zxc_min_level(void) { return ZXC_LEVEL_FASTEST; }
int zxc_max_level(void) { return ZXC_LEVEL_ULTRA; }
int zxc_default_level(void) { return ZXC_LEVEL_DEFAULT; }
const char* zxc_version_string(void) { return ZXC_LIB_VERSION_STR; }
size_t zxc_compress_opts_size(void) { return sizeof(zxc_compress_opts_t); }
size_t zxc_decompress_opts_size(void) { return sizeof(zxc_decompress_opts_t); }

/* src/lib/zxc_common.c:850-862: header + per-4KiB-block overhead (8 hdr + 4 cksum + 68 fmt)
 * + input + EOF block + SEK header + 4 B/block + footer */
uint64_t zxc_compress_bound(const size_t input_size) {
    if (input_size > (SIZE_MAX - (SIZE_MAX >> 8))) return 0;
    uint64_t n = ((uint64_t)input_size + ZXC_BLOCK_SIZE_MIN - 1) / ZXC_BLOCK_SIZE_MIN;
    if (n == 0) n = 1;
    return ZXC_FILE_HEADER_SIZE + n * (8 + 4 + 68) + (uint64_t)input_size + 8 + 8 + n * 4 + ZXC_FILE_FOOTER_SIZE;
}

/* ------------------------------------------------- dictionary id (host side) */
/* zxc_dict_id (src/lib/zxc_dict.c:35-45) = rapidhash v3 of the content folded to 32 bits, chained
 * over the 128-byte shared table when there is one. rapidhash restated from its published
 * algorithm (the reference vendors it under src/lib/vendors/rapidhash.h). Header-level work: it
 * decides DICT_MISMATCH before anything is sent to the GPU. */
static const uint64_t RH_S[8] = {0x2d358dccaa6c78a5ull, 0x8bb84b93962eacc9ull, 0x4b33a62ed433d4a3ull,
                                 0x4d5a2da51de1aa47ull, 0xa0761d6478bd642full, 0xe7037ed1a0b428dbull,
                                 0x90ed1765281c388cull, 0xaaaaaaaaaaaaaaaaull};
static uint64_t rh_mix(uint64_t a, uint64_t b) {
    const __uint128_t r = (__uint128_t)a * b;
    return (uint64_t)r ^ (uint64_t)(r >> 64);
}
static uint64_t host_rapidhash(const uint8_t* p, size_t len, uint64_t seed) {
    uint64_t a = 0, b = 0;
    size_t i = len;
    seed ^= rh_mix(seed ^ RH_S[2], RH_S[1]);
    if (len <= 16) {
        if (len >= 4) {
            seed ^= len;
            if (len >= 8) { a = rd64(p); b = rd64(p + len - 8); }
            else { a = rd32(p); b = rd32(p + len - 4); }
        } else if (len > 0) {
            a = ((uint64_t)p[0] << 45) | p[len - 1];
            b = p[len >> 1];
        }
    } else {
        if (len > 112) {
            uint64_t s[7];
            for (int k = 0; k < 7; k++) s[k] = seed;
            do {
                for (int k = 0; k < 7; k++) s[k] = rh_mix(rd64(p + 16 * k) ^ RH_S[k], rd64(p + 16 * k + 8) ^ s[k]);
                p += 112;
                i -= 112;
            } while (i > 112);
            seed = s[0] ^ s[1] ^ s[2] ^ s[3] ^ s[4] ^ s[5] ^ s[6];
        }
        static const int sel[6] = {2, 2, 1, 1, 2, 1};
        for (int k = 0; k < 6 && i > (size_t)(16 * (k + 1)); k++)
            seed = rh_mix(rd64(p + 16 * k) ^ RH_S[sel[k]], rd64(p + 16 * k + 8) ^ seed);
        a = rd64(p + i - 16) ^ i;
        b = rd64(p + i - 8);
    }
    a ^= RH_S[1];
    b ^= seed;
    const __uint128_t r = (__uint128_t)a * b;
    return rh_mix((uint64_t)r ^ RH_S[7], (uint64_t)(r >> 64) ^ RH_S[1] ^ i);
}
static uint32_t dict_id_of(const uint8_t* dict, size_t n, const uint8_t* huf) {
    if (!dict || n == 0) return 0;
    uint64_t h = host_rapidhash(dict, n, 0);
    const uint32_t base = (uint32_t)(h ^ (h >> 32));
    if (!huf) return base;
    h = host_rapidhash(huf, ZXC_HUF_TABLE_SIZE, base);
    return (uint32_t)(h ^ (h >> 32));
}

/* .zxd container (reference src/lib/zxc_dict.c:35-205): 16-byte header {magic, version, flags, content size u16,
 * dict_id u32, reserved u16, CRC16 over the header with bytes 12-15 zeroed} + content + 128-byte table */
#define DICT_MAGIC 0x9CB0D1C7u
#define DICT_VERSION 1
#define DICT_HDR 16
static uint16_t hdr_hash16(const uint8_t* p);

uint32_t zxc_dict_id(const void* dict, size_t dict_size, const void* huf_lengths) {
    return dict_id_of((const uint8_t*)dict, dict_size, (const uint8_t*)huf_lengths);
}
uint32_t zxc_dict_get_id(const void* buf, const size_t buf_size) {
    if (!buf || buf_size < DICT_HDR) return 0;
    const uint8_t* p = (const uint8_t*)buf;
    return rd32(p) == DICT_MAGIC ? rd32(p + 8) : 0;
}
size_t zxc_dict_save_bound(const size_t content_size) { return DICT_HDR + content_size + ZXC_HUF_TABLE_SIZE; }
int64_t zxc_dict_save(const void* content, const size_t content_size, const void* huf_lengths, void* buf,
                      const size_t buf_capacity) {
    if (!content || content_size == 0 || !huf_lengths) return ZXC_ERROR_NULL_INPUT;
    if (content_size > ZXC_DICT_SIZE_MAX) return ZXC_ERROR_DICT_TOO_LARGE;
    const size_t total = zxc_dict_save_bound(content_size);
    if (buf_capacity < total) return ZXC_ERROR_DST_TOO_SMALL;
    uint8_t* d = (uint8_t*)buf;
    wr32(d, DICT_MAGIC);
    d[4] = DICT_VERSION;
    d[5] = 0;
    d[6] = (uint8_t)content_size;
    d[7] = (uint8_t)(content_size >> 8);
    wr32(d + 8, dict_id_of((const uint8_t*)content, content_size, (const uint8_t*)huf_lengths));
    wr32(d + 12, 0);
    const uint16_t crc = hdr_hash16(d);
    d[14] = (uint8_t)crc;
    d[15] = (uint8_t)(crc >> 8);
    memcpy(d + DICT_HDR, content, content_size);
    memcpy(d + DICT_HDR + content_size, huf_lengths, ZXC_HUF_TABLE_SIZE);
    return (int64_t)total;
}
int zxc_dict_load(const void* buf, const size_t buf_size, const void** content_out, size_t* content_size_out,
                  const void** huf_out, uint32_t* dict_id_out) {
    if (!buf || !content_out || !content_size_out) return ZXC_ERROR_NULL_INPUT;
    if (buf_size < DICT_HDR) return ZXC_ERROR_SRC_TOO_SMALL;
    const uint8_t* src = (const uint8_t*)buf;
    if (rd32(src) != DICT_MAGIC) return ZXC_ERROR_BAD_MAGIC;
    if (src[4] != DICT_VERSION) return ZXC_ERROR_BAD_VERSION;
    const size_t n = rd16(src + 6);
    if (n == 0) return ZXC_ERROR_CORRUPT_DATA;
    if (buf_size < DICT_HDR + n + ZXC_HUF_TABLE_SIZE) return ZXC_ERROR_SRC_TOO_SMALL;
    uint8_t t[DICT_HDR];
    memcpy(t, src, DICT_HDR);
    t[12] = t[13] = t[14] = t[15] = 0;
    if (rd16(src + 14) != hdr_hash16(t)) return ZXC_ERROR_BAD_HEADER;
    const uint8_t* content = src + DICT_HDR;
    const uint8_t* huf = content + n;
    const uint32_t id = dict_id_of(content, n, huf);
    if (rd32(src + 8) != id) return ZXC_ERROR_BAD_CHECKSUM;
    *content_out = content;
    *content_size_out = n;
    if (huf_out) *huf_out = huf;
    if (dict_id_out) *dict_id_out = id;
    return ZXC_OK;
}
const void* zxc_dict_huf(const void* buf, const size_t buf_size) {
    if (!buf || buf_size < DICT_HDR) return NULL;
    const uint8_t* src = (const uint8_t*)buf;
    if (rd32(src) != DICT_MAGIC || src[4] != DICT_VERSION) return NULL;
    const size_t n = rd16(src + 6);
    if (n == 0 || buf_size < DICT_HDR + n + ZXC_HUF_TABLE_SIZE) return NULL;
    return src + DICT_HDR + n;
}

/* ------------------------------------------------------------- containers */
static int read_file_header(const uint8_t* src, size_t n, uint32_t* block_size, int* has_checksum,
                            uint32_t* dict_id) {
    if (n < ZXC_FILE_HEADER_SIZE) return ZXC_ERROR_SRC_TOO_SMALL;
    if (rd32(src) != MAGIC) return ZXC_ERROR_BAD_MAGIC;
    if (src[4] != FORMAT_VERSION) return ZXC_ERROR_BAD_VERSION;
    uint8_t t[16];
    memcpy(t, src, 16);
    t[14] = t[15] = 0;
    if (rd16(src + 14) != hdr_hash16(t) || (src[6] & 0x0F) != 0) return ZXC_ERROR_BAD_HEADER;
    if (src[5] < ZXC_BLOCK_SIZE_MIN_LOG2 || src[5] > ZXC_BLOCK_SIZE_MAX_LOG2) return ZXC_ERROR_BAD_BLOCK_SIZE;
    *block_size = 1u << src[5];
    *has_checksum = (src[6] & 0x80) ? 1 : 0;
    *dict_id = (src[6] & 0x40) ? rd32(src + 7) : 0;
    return ZXC_OK;
}

static int read_block_header(const uint8_t* src, size_t n, uint8_t* type, uint32_t* comp_size) {
    if (n < BLK_HDR) return ZXC_ERROR_SRC_TOO_SMALL;
    uint8_t t[8];
    memcpy(t, src, 8);
    t[7] = 0;
    if (src[7] != hdr_hash8(t)) return ZXC_ERROR_BAD_HEADER;
    *type = src[0];
    *comp_size = rd32(src + 3);
    return ZXC_OK;
}

uint64_t zxc_get_decompressed_size(const void* src, const size_t src_size) {
    if (!src || src_size < ZXC_FILE_HEADER_SIZE + ZXC_FILE_FOOTER_SIZE) return 0;
    const uint8_t* p = (const uint8_t*)src;
    uint32_t bs, did;
    int cs;
    if (read_file_header(p, src_size, &bs, &cs, &did) != ZXC_OK) return 0;
    const uint64_t dsize = rd64(p + src_size - ZXC_FILE_FOOTER_SIZE);
    /* plausibility cap: each block costs >= 8 compressed bytes (zxc_dispatch.c:1021-1027) */
    const uint64_t need = dsize / bs + (dsize % bs != 0);
    return need <= (uint64_t)(src_size / BLK_HDR) ? dsize : 0;
}

/* ------------------------------------------------------- device round trip */
/* hidden entry points of zxc_hip_shim.hip */
int zxc_hip_decode_blocks(const void* d_comp, const zxc_dev_job_t* d_jobs, uint32_t n_jobs, void* d_out, int32_t* d_status,
                          uint32_t block_size, int verify_trailer, const void* d_dict, uint32_t dict_size,
                          const void* d_dict_huf, uint32_t cap_override, void* stream);
int zxc_hip_current_device(void);

/* hidden entry points of zxc_hip_shim.hip: a worker's own stream */
int zxc_hip_stream_create(void** stream_out);
void zxc_hip_stream_destroy(void* stream);
int zxc_hip_memcpy_h2d_async(void* d_dst, const void* h_src, size_t bytes, void* stream);
int zxc_hip_memcpy_d2h_async(void* h_dst, const void* d_src, size_t bytes, void* stream);

/* Staging arenas: the device buffers of the host API live across calls (grown on demand), so a call costs its copies
 * and its launch, not four hipMalloc + hipFree (which synchronise the device). One call at a time uses an arena; its
 * mutex is held from run_jobs() until dev_bufs_free(). Every device has ARENA_SUBS of them: sub 0 serves the
 * single-threaded API, the workers of zxc_seekable_decompress_range_mt take one each (two workers on one device run on
 * two streams side by side). Every caller batches its work (HOST_BATCH_BYTES of output slots per launch), so an arena
 * stays within a few hundred MiB; a buffer that grew beyond ARENA_KEEP_MAX is given back when the call ends, and
 * zxc_mi355x_release_cached() frees everything that is not in use. */
typedef struct { void* p; size_t cap; } dbuf_t;
enum { AR_COMP = 0, AR_JOBS, AR_OUT, AR_STATUS, AR_DICT, AR_N };
typedef struct {
    pthread_mutex_t mu;
    dbuf_t buf[AR_N];
} arena_t;
#define HOST_MAX_DEVICES 16
#define ARENA_SUBS 4
#define ARENA_KEEP_MAX ((size_t)768 << 20)
static arena_t g_arena[HOST_MAX_DEVICES][ARENA_SUBS];
static pthread_once_t g_arena_once = PTHREAD_ONCE_INIT;
static void arena_init_all(void) {
    for (int i = 0; i < HOST_MAX_DEVICES; i++)
        for (int k = 0; k < ARENA_SUBS; k++) pthread_mutex_init(&g_arena[i][k].mu, NULL);
}
static void* arena_reserve(arena_t* a, int which, size_t need) {
    dbuf_t* d = &a->buf[which];
    if (d->cap < need) {
        zxc_mi355x_free(d->p);
        d->cap = 0;
        size_t want = need + (need >> 3) + 4096; /* a little slack: slowly growing callers do not realloc every time */
        d->p = zxc_mi355x_malloc(want);
        if (!d->p) { want = need; d->p = zxc_mi355x_malloc(want); }
        if (d->p) d->cap = want;
    }
    return d->p;
}
/* internal to the library (called by zxc_mi355x_release_cached in the shim): free every arena nobody holds */
void zxc_host_release_arenas(void) {
    pthread_once(&g_arena_once, arena_init_all);
    const int cur = zxc_hip_current_device();
    for (int dev = 0; dev < HOST_MAX_DEVICES; dev++)
        for (int k = 0; k < ARENA_SUBS; k++) {
            arena_t* a = &g_arena[dev][k];
            if (pthread_mutex_trylock(&a->mu) != 0) continue; /* in use: its call trims it when it ends */
            int any = 0;
            for (int w = 0; w < AR_N; w++) any |= a->buf[w].p != NULL;
            if (any && zxc_mi355x_set_device(dev) == ZXC_OK)
                for (int w = 0; w < AR_N; w++) { zxc_mi355x_free(a->buf[w].p); a->buf[w].p = NULL; a->buf[w].cap = 0; }
            pthread_mutex_unlock(&a->mu);
        }
    if (cur >= 0) (void)zxc_mi355x_set_device(cur);
}

/* Decode `n` jobs whose compressed bytes are h_comp[0..comp_bytes) on the host.
 * Output slot i is at jobs[i].out_off. Leaves statuses in h_status and, on success of the
 * launch, the decoded slots in b->d_out (device memory of the arena: valid until dev_bufs_free(b)). */
typedef struct {
    void* d_comp;
    void* d_jobs;
    void* d_out;
    void* d_status;
    void* d_dict; /* [dict content | 128-byte shared table] or NULL */
    arena_t* held;
    void* stream; /* the stream the batch ran on (NULL: the device's default stream) */
} dev_bufs_t;

typedef struct {
    const uint8_t* dict;
    size_t dict_size;
    const uint8_t* dict_huf;
} dict_ref_t;

static void dev_bufs_free(dev_bufs_t* b) {
    arena_t* a = b->held;
    memset(b, 0, sizeof(*b));
    if (a) {
        for (int w = 0; w < AR_N; w++)
            if (a->buf[w].cap > ARENA_KEEP_MAX) { zxc_mi355x_free(a->buf[w].p); a->buf[w].p = NULL; a->buf[w].cap = 0; }
        pthread_mutex_unlock(&a->mu);
    }
}
/* decoded bytes of the batch back to the host, on the batch's stream */
static int dev_bufs_read(const dev_bufs_t* b, void* h_dst, size_t d_off, size_t bytes) {
    if (bytes == 0) return ZXC_OK;
    if (!b->stream) return zxc_mi355x_memcpy_d2h(h_dst, (const uint8_t*)b->d_out + d_off, bytes);
    const int rc = zxc_hip_memcpy_d2h_async(h_dst, (const uint8_t*)b->d_out + d_off, bytes, b->stream);
    return rc == ZXC_OK ? zxc_mi355x_synchronize(b->stream) : rc;
}

/* Stage 1 of a batch: take the arena (dev, sub), reserve its buffers and upload the compressed bytes, the job table and the
 * dictionary. `stream` NULL: plain (synchronous) copies; else copies on that stream, complete when this returns. On failure the
 * arena is released. The two stages are separate so that zxc_decompress can upload batch i+1 from a helper thread while batch i's
 * output goes back to the host (PCIe is full duplex; one direction at a time left a third of the link idle). */
static int jobs_upload(const uint8_t* h_comp, size_t comp_bytes, const zxc_dev_job_t* jobs, uint32_t n, size_t out_bytes,
                       dev_bufs_t* b, const dict_ref_t* dr, int sub, void* stream, arena_t* locked) {
    /* `locked`: the arena to use, already locked by the thread that will release it (a mutex is released by the thread that took
     * it: zxc_decompress takes the helper's arena itself); a failure then leaves the release to that caller too */
    memset(b, 0, sizeof(*b));
    b->held = locked;
    if (zxc_mi355x_device_count() <= 0) return ZXC_ERROR_GPU_UNAVAILABLE;
    const int dev = zxc_hip_current_device();
    if (dev < 0 || dev >= HOST_MAX_DEVICES || sub < 0 || sub >= ARENA_SUBS) return ZXC_ERROR_GPU_UNAVAILABLE;
    pthread_once(&g_arena_once, arena_init_all);
    arena_t* a = locked ? locked : &g_arena[dev][sub];
    if (!locked) pthread_mutex_lock(&a->mu);
    b->held = a;
    b->stream = stream;
    /* +64: the kernel's 16-byte literal / extras reads may run past the last block */
    b->d_comp = arena_reserve(a, AR_COMP, comp_bytes + 64);
    b->d_jobs = arena_reserve(a, AR_JOBS, (size_t)n * sizeof(zxc_dev_job_t));
    b->d_out = arena_reserve(a, AR_OUT, out_bytes + 64);
    b->d_status = arena_reserve(a, AR_STATUS, (size_t)n * sizeof(int32_t));
    int rc = ZXC_ERROR_MEMORY;
    if (b->d_comp && b->d_jobs && b->d_out && b->d_status) {
        rc = stream ? zxc_hip_memcpy_h2d_async(b->d_comp, h_comp, comp_bytes, stream) : zxc_mi355x_memcpy_h2d(b->d_comp, h_comp, comp_bytes);
        if (rc == ZXC_OK)
            rc = stream ? zxc_hip_memcpy_h2d_async(b->d_jobs, jobs, (size_t)n * sizeof(zxc_dev_job_t), stream)
                        : zxc_mi355x_memcpy_h2d(b->d_jobs, jobs, (size_t)n * sizeof(zxc_dev_job_t));
        if (rc == ZXC_OK && stream) rc = zxc_mi355x_synchronize(stream);
        if (rc == ZXC_OK && dr && dr->dict_size) {
            b->d_dict = arena_reserve(a, AR_DICT, dr->dict_size + ZXC_HUF_TABLE_SIZE + 64);
            if (!b->d_dict) rc = ZXC_ERROR_MEMORY;
            if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_h2d(b->d_dict, dr->dict, dr->dict_size); /* (the dictionary goes up with plain copies) */
            if (rc == ZXC_OK && dr->dict_huf)
                rc = zxc_mi355x_memcpy_h2d((uint8_t*)b->d_dict + dr->dict_size, dr->dict_huf, ZXC_HUF_TABLE_SIZE);
        }
    }
    if (rc != ZXC_OK && !locked) dev_bufs_free(b);
    return rc;
}
/* Stage 2: the launch over an uploaded batch and its statuses back on the host (synchronises). */
static int jobs_execute(dev_bufs_t* b, uint32_t n, uint32_t block_size, uint32_t cap_override, int verify_trailer,
                        int32_t* h_status, const dict_ref_t* dr, void* stream) {
    b->stream = stream;
    int rc = zxc_hip_decode_blocks(b->d_comp, (const zxc_dev_job_t*)b->d_jobs, n, b->d_out, (int32_t*)b->d_status,
                                   block_size, verify_trailer, b->d_dict, b->d_dict ? (uint32_t)dr->dict_size : 0u,
                                   (b->d_dict && dr->dict_huf) ? (uint8_t*)b->d_dict + dr->dict_size : NULL,
                                   cap_override, stream);
    if (rc == ZXC_OK && stream) {
        rc = zxc_hip_memcpy_d2h_async(h_status, b->d_status, (size_t)n * sizeof(int32_t), stream);
        if (rc == ZXC_OK) rc = zxc_mi355x_synchronize(stream);
    } else if (rc == ZXC_OK) {
        rc = zxc_mi355x_synchronize(NULL);
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(h_status, b->d_status, (size_t)n * sizeof(int32_t));
    }
    if (rc != ZXC_OK) dev_bufs_free(b);
    return rc;
}
static int run_jobs_on(const uint8_t* h_comp, size_t comp_bytes, const zxc_dev_job_t* jobs, uint32_t n,
                       size_t out_bytes, uint32_t block_size, uint32_t cap_override, int verify_trailer,
                       int32_t* h_status, dev_bufs_t* b, const dict_ref_t* dr, int sub, void* stream) {
    const int rc = jobs_upload(h_comp, comp_bytes, jobs, n, out_bytes, b, dr, sub, stream, NULL);
    return rc == ZXC_OK ? jobs_execute(b, n, block_size, cap_override, verify_trailer, h_status, dr, stream) : rc;
}

static int run_jobs_cap(const uint8_t* h_comp, size_t comp_bytes, const zxc_dev_job_t* jobs, uint32_t n,
                        size_t out_bytes, uint32_t block_size, uint32_t cap_override, int verify_trailer,
                        int32_t* h_status, dev_bufs_t* b, const dict_ref_t* dr) {
    return run_jobs_on(h_comp, comp_bytes, jobs, n, out_bytes, block_size, cap_override, verify_trailer, h_status, b, dr, 0, NULL);
}

static int run_jobs(const uint8_t* h_comp, size_t comp_bytes, const zxc_dev_job_t* jobs, uint32_t n,
                    size_t out_bytes, uint32_t block_size, int verify_trailer, int32_t* h_status,
                    dev_bufs_t* b, const dict_ref_t* dr) {
    return run_jobs_cap(h_comp, comp_bytes, jobs, n, out_bytes, block_size, 0u, verify_trailer, h_status, b, dr);
}

/* ------------------------------------------------------------ zxc_decompress */
typedef struct {
    const uint8_t* h_comp;
    size_t comp_bytes;
    const zxc_dev_job_t* jobs;
    uint32_t n;
    size_t out_bytes;
    dev_bufs_t* b;
    const dict_ref_t* dr;
    int sub;
    void* stream;
    arena_t* arena; /* locked by the thread that started the helper */
    int device, rc;
    uint32_t block_size; /* the launch over the uploaded batch, on the helper's stream, and its statuses */
    int verify;
    int32_t* st;
} host_upload_t;
static void* host_upload_main(void* arg) {
    host_upload_t* u = (host_upload_t*)arg;
    u->rc = zxc_mi355x_set_device(u->device);
    if (u->rc == ZXC_OK) u->rc = jobs_upload(u->h_comp, u->comp_bytes, u->jobs, u->n, u->out_bytes, u->b, u->dr, u->sub, u->stream, u->arena);
    if (u->rc == ZXC_OK) { /* (jobs_execute releases the arena when it fails: not ours to release here — keep the handle for the caller) */
        dev_bufs_t tmp = *u->b;
        tmp.held = NULL;
        u->rc = jobs_execute(&tmp, u->n, u->block_size, 0u, u->verify, u->st, u->dr, u->stream);
    }
    return NULL;
}

int64_t zxc_decompress(const void* src_v, const size_t src_size, void* dst_v, const size_t dst_capacity,
                       const zxc_decompress_opts_t* opts) {
    const uint8_t* src = (const uint8_t*)src_v;
    uint8_t* dst = (uint8_t*)dst_v;
    if (!src || (!dst && dst_capacity != 0)) return ZXC_ERROR_NULL_INPUT;
    if (src_size < ZXC_FILE_HEADER_SIZE + ZXC_FILE_FOOTER_SIZE) return ZXC_ERROR_SRC_TOO_SMALL;
    if (!dst || dst_capacity == 0) { /* empty-frame probe, zxc_dispatch.c:848-853 */
        if (rd32(src) != MAGIC) return ZXC_ERROR_BAD_MAGIC;
        return rd64(src + src_size - ZXC_FILE_FOOTER_SIZE) == 0 ? 0 : (int64_t)ZXC_ERROR_DST_TOO_SMALL;
    }
    uint32_t block_size, dict_id;
    int file_ck;
    const int hrc = read_file_header(src, src_size, &block_size, &file_ck, &dict_id);
    if (hrc != ZXC_OK) return hrc;
    const int verify = file_ck && opts && opts->checksum_enabled;
    const uint8_t* dict = opts ? (const uint8_t*)opts->dict : NULL;
    const size_t dict_size = (opts && opts->dict) ? opts->dict_size : 0;
    const uint8_t* dict_huf = (opts && opts->dict) ? (const uint8_t*)opts->dict_huf : NULL;
    if (dict_id != 0) { /* zxc_dispatch.c:883-892 */
        if (!dict || dict_size == 0) return ZXC_ERROR_DICT_REQUIRED;
        if (dict_id_of(dict, dict_size, dict_huf) != dict_id) return ZXC_ERROR_DICT_MISMATCH;
    }
    if (dict_size > ZXC_DICT_SIZE_MAX) return ZXC_ERROR_DICT_TOO_LARGE;
    const dict_ref_t dr = {dict, dict_size, dict_huf};

    /* The 8-byte block headers are walked on the host into a job table, one bounded batch at a time
     * (at most HOST_BATCH_BYTES of output slots per launch): device memory is O(batch), not a function
     * of the untrusted block count, and decoding stops at the first failing or overflowing block like
     * the reference's sequential loop (zxc_dispatch.c:912-1001). A problem found at block k is only
     * reported if blocks 0..k-1 all decode (first failure in stream order wins). */
    size_t batch_bytes = FRAME_BATCH_BYTES;
    { const char* e = getenv("ZXC_MI355X_FRAME_BATCH_MIB"); if (e && atoi(e) >= 1 && atoi(e) <= 1024) batch_bytes = (size_t)atoi(e) << 20; }
    uint32_t batch_blocks = (uint32_t)(batch_bytes / block_size);
    if (batch_blocks < 16u) batch_blocks = 16u;
    /* Two batches in flight: while batch i's decoded bytes go back to the host on this thread, a helper thread uploads batch
     * i+1 (its own arena, its own stream) — the link carries both directions at once. W[k]: the walked batch in slot k. */
    struct { zxc_dev_job_t* jobs; int32_t* st; uint32_t n; size_t span0, span; int last, uploaded; dev_bufs_t b; } W[2];
    memset(W, 0, sizeof W);
    for (int k = 0; k < 2; k++) {
        W[k].jobs = (zxc_dev_job_t*)malloc((size_t)batch_blocks * sizeof(zxc_dev_job_t));
        W[k].st = (int32_t*)malloc((size_t)batch_blocks * sizeof(int32_t));
    }
    if (!W[0].jobs || !W[1].jobs || !W[0].st || !W[1].st) {
        for (int k = 0; k < 2; k++) { free(W[k].jobs); free(W[k].st); }
        return ZXC_ERROR_MEMORY;
    }
    size_t ip = ZXC_FILE_HEADER_SIZE;
    int tail_err = 0;         /* error to report after all queued blocks succeed */
    uint32_t global_hash = 0; /* rotl1-xor fold of the stored per-block checksums (zxc_internal.h:1390-1393) */
    int saw_eof = 0, done = 0;
    int64_t ret = 0;
    size_t total = 0;
    const uint32_t slot = (block_size + TAIL_PAD + 15u) & ~15u;
    void* up_stream = NULL;   /* the helper's stream, created with the first second batch */
    /* the 8-byte block headers of the next batch -> W[k] (host only) */
#define WALK_BATCH(k)                                                                                                      \
    do {                                                                                                                   \
        zxc_dev_job_t* jobs = W[k].jobs;                                                                                   \
        uint32_t n = 0;                                                                                                    \
        const size_t span0 = ip;                                                                                           \
        while (!done && n < batch_blocks) {                                                                                \
            if (ip >= src_size) { done = 1; break; }                                                                       \
            const size_t rem = src_size - ip;                                                                              \
            uint8_t type;                                                                                                  \
            uint32_t csz;                                                                                                  \
            if (read_block_header(src + ip, rem, &type, &csz) != ZXC_OK) { tail_err = ZXC_ERROR_BAD_HEADER; done = 1; break; } \
            if (type == BLK_EOF) {                                                                                         \
                if (csz != 0) tail_err = ZXC_ERROR_BAD_HEADER;                                                             \
                saw_eof = 1;                                                                                               \
                done = 1;                                                                                                  \
                break;                                                                                                     \
            }                                                                                                              \
            const uint64_t phys = (uint64_t)BLK_HDR + csz + (file_ck ? 4u : 0u);                                           \
            jobs[n].comp_off = ip - span0;                                                                                 \
            /* the wrapper sees "all remaining bytes"; any size >= the physical block is equivalent */                    \
            { const uint64_t cs = phys < rem ? phys : rem; jobs[n].comp_size = cs > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cs; } \
            jobs[n].out_off = (uint64_t)n * block_size;                                                                    \
            jobs[n].out_len = block_size;                                                                                  \
            n++;                                                                                                           \
            if (verify && phys <= rem)                                                                                     \
                global_hash = ((global_hash << 1) | (global_hash >> 31)) ^ rd32(src + ip + BLK_HDR + csz);                 \
            if (phys >= rem) { ip = src_size; done = 1; break; }                                                           \
            ip += (size_t)phys;                                                                                            \
        }                                                                                                                  \
        W[k].n = n;                                                                                                        \
        W[k].span0 = span0;                                                                                                \
        W[k].span = ip - span0;                                                                                            \
        W[k].last = done;                                                                                                  \
    } while (0)
    int cur = 0;
    WALK_BATCH(0);
    const int my_dev = zxc_hip_current_device();
    while (W[cur].n && ret == 0) {
        const uint32_t n = W[cur].n;
        zxc_dev_job_t* jobs = W[cur].jobs;
        const size_t span0 = W[cur].span0, span = W[cur].span;
        int32_t* st = W[cur].st;
        int rc = ZXC_OK;
        dev_bufs_t b;
        if (!W[cur].uploaded) { /* the first batch, or one whose arena was busy when it was walked: this thread holds nothing here */
            rc = jobs_upload(src + span0, span, jobs, n, (size_t)n * block_size, &W[cur].b, &dr, cur, NULL, NULL);
            b = W[cur].b;
            W[cur].n = 0; /* (from here on `b` owns the arena) */
            if (rc == ZXC_OK) rc = jobs_execute(&b, n, block_size, 0u, verify, st, &dr, NULL);
        } else { /* uploaded AND decoded by the helper, statuses in st */
            b = W[cur].b;
            W[cur].n = 0;
        }
        if (rc != ZXC_OK) { ret = rc; break; }
        /* the next batch: walked here, uploaded AND decoded by the helper (its own arena and stream) while this batch's output is
         * copied back: three stages overlap — a launch costs ~0.5 ms whatever its size, serial per batch that was 0.6 ms of every
         * batch (profiles/r4c_host_api.log) */
        const int nxt = cur ^ 1;
        host_upload_t up;
        memset(&up, 0, sizeof up);
        pthread_t up_thread;
        int up_live = 0;
        W[nxt].n = 0;
        W[nxt].uploaded = 0;
        if (!done) {
            WALK_BATCH(nxt);
            /* the other arena, if nobody holds it (never WAIT for a second arena while holding one: two such callers would
             * wait for each other); else the batch is uploaded in series when this one is done */
            arena_t* na = (W[nxt].n && my_dev >= 0 && my_dev < HOST_MAX_DEVICES) ? &g_arena[my_dev][nxt] : NULL;
            if (na && !up_stream && zxc_hip_stream_create(&up_stream) != ZXC_OK) up_stream = NULL;
            if (na && up_stream && pthread_mutex_trylock(&na->mu) == 0) {
                up = (host_upload_t){src + W[nxt].span0, W[nxt].span, W[nxt].jobs, W[nxt].n, (size_t)W[nxt].n * block_size, &W[nxt].b, &dr,
                                     nxt, up_stream, na, my_dev, ZXC_OK, block_size, verify, W[nxt].st};
                up_live = pthread_create(&up_thread, NULL, host_upload_main, &up) == 0;
                if (!up_live) host_upload_main(&up); /* (no thread: upload it here, in series) */
                W[nxt].uploaded = 1;
            }
        }
        /* sequential semantics: first failing block wins; sizes accumulate in order. A block that is not the
         * frame's last and decodes to another size than block_size makes the frame irregular (legal, never
         * produced by the reference encoder): blocks no longer sit back to back in the slot layout. */
        int regular = 1;
        size_t batch_total = 0;
        for (uint32_t i = 0; i < n; i++) {
            if (st[i] < 0) { ret = st[i]; break; }
            if ((size_t)st[i] > dst_capacity - total - batch_total) { ret = ZXC_ERROR_DST_TOO_SMALL; break; }
            if ((uint32_t)st[i] != block_size && !(W[cur].last && i + 1 == n)) regular = 0;
            if ((uint32_t)st[i] > block_size) regular = 0;
            batch_total += (size_t)st[i];
        }
        if (ret == 0 && regular) {
            const int crc = zxc_mi355x_memcpy_d2h(dst + total, b.d_out, batch_total);
            if (crc != ZXC_OK) ret = crc;
        }
        if (up_live) pthread_join(up_thread, NULL);
        if (W[nxt].uploaded && up.rc != ZXC_OK) { /* a failed upload: its arena is ours to release */
            if (ret == 0) ret = up.rc;
            W[nxt].b.held = up.arena;
            dev_bufs_free(&W[nxt].b);
            W[nxt].n = 0;
            W[nxt].uploaded = 0;
        }
        if (ret == 0 && !regular) {
            /* decoded sizes are a property of the blocks alone: re-run this batch with one cap-sized slot
             * per block and gather. (The prefetched batch gives its arena back first — this thread is about to wait for one —
             * and goes up again, in series, when its turn comes.) */
            if (W[nxt].uploaded) { dev_bufs_free(&W[nxt].b); W[nxt].uploaded = 0; }
            dev_bufs_free(&b);
            for (uint32_t i = 0; i < n; i++) { jobs[i].out_off = (uint64_t)i * slot; jobs[i].out_len = slot; }
            rc = run_jobs_on(src + span0, span, jobs, n, (size_t)n * slot, block_size, 0u, verify, st, &b, &dr, cur, NULL);
            if (rc != ZXC_OK) { ret = rc; memset(&b, 0, sizeof b); }
            size_t op = total;
            for (uint32_t i = 0; i < n && rc == ZXC_OK; i++) {
                /* the second run must reproduce the first one's verdicts: a status is never trusted as a copy length */
                if (st[i] < 0) { rc = st[i]; break; }
                if ((uint32_t)st[i] > slot || (size_t)st[i] > dst_capacity - op) { rc = ZXC_ERROR_CORRUPT_DATA; break; }
                rc = zxc_mi355x_memcpy_d2h(dst + op, (const uint8_t*)b.d_out + (size_t)i * slot, (size_t)st[i]);
                op += (size_t)st[i];
            }
            if (rc != ZXC_OK && ret == 0) ret = rc;
        }
        dev_bufs_free(&b);
        total += batch_total;
        cur = nxt;
    }
#undef WALK_BATCH
    for (int k = 0; k < 2; k++)
        if (W[k].n && W[k].uploaded) dev_bufs_free(&W[k].b); /* (an uploaded batch that is not going to run: give its arena back) */
    zxc_hip_stream_destroy(up_stream);
    for (int k = 0; k < 2; k++) { free(W[k].jobs); free(W[k].st); }
    if (ret < 0) return ret;
    if (tail_err) return tail_err;
    if (saw_eof) { /* footer: stored size must equal what was produced (zxc_dispatch.c:936-943) */
        const uint8_t* footer = src + src_size - ZXC_FILE_FOOTER_SIZE;
        if (rd64(footer) != (uint64_t)total) return ZXC_ERROR_CORRUPT_DATA;
        if (verify && rd32(footer + 8) != global_hash) return ZXC_ERROR_BAD_CHECKSUM; /* :945-952 */
    }
    return (int64_t)total;
}

/* ------------------------------------------------ in-place decode (one buffer) */
/* reference include/zxc_buffer.h:155-181, impl src/lib/zxc_dispatch.c:1056-1185: the archive sits flush-right in the
 * buffer, the output grows from its start; the bound keeps the write cursor behind the read cursor. Here blocks are
 * uploaded batch by batch before their output comes back, and a batch's output ends where the sequential decoder's
 * would, so the same bound holds. */
static int inplace_probe(const uint8_t* comp, size_t comp_size, uint64_t* dsize, uint64_t* margin, uint64_t* floor_) {
    if (rd32(comp) != MAGIC) return ZXC_ERROR_BAD_MAGIC;
    uint32_t bs, did;
    int ck;
    if (read_file_header(comp, comp_size, &bs, &ck, &did) != ZXC_OK) return ZXC_ERROR_BAD_HEADER;
    const uint64_t d = rd64(comp + comp_size - ZXC_FILE_FOOTER_SIZE);
    const uint64_t need = d / bs + (d % bs != 0);
    if (need > (uint64_t)(comp_size / BLK_HDR)) return ZXC_ERROR_CORRUPT_DATA;
    const uint64_t nblocks = (d + bs - 1) / bs;
    const uint64_t per_block = BLK_HDR + (ck ? 4u : 0u);
    const uint64_t trailing = BLK_HDR + (BLK_HDR + nblocks * 4u) + ZXC_FILE_FOOTER_SIZE;
    *dsize = d;
    *margin = (uint64_t)bs + nblocks * per_block + trailing + TAIL_PAD;
    *floor_ = (uint64_t)bs + TAIL_PAD;
    return ZXC_OK;
}
size_t zxc_decompress_inplace_bound(const void* src, const size_t src_size) {
    if (!src || src_size < ZXC_FILE_HEADER_SIZE + ZXC_FILE_FOOTER_SIZE) return 0;
    uint64_t dsize = 0, margin = 0, fl = 0;
    if (inplace_probe((const uint8_t*)src, src_size, &dsize, &margin, &fl) != ZXC_OK) return 0;
    if (margin > (uint64_t)SIZE_MAX || dsize > (uint64_t)SIZE_MAX - margin) return 0;
    if (fl > (uint64_t)SIZE_MAX - (uint64_t)src_size) return 0;
    const uint64_t a = dsize + margin, b = (uint64_t)src_size + fl;
    return (size_t)(a > b ? a : b);
}
int64_t zxc_decompress_inplace(void* buffer, const size_t buffer_capacity, const size_t comp_size,
                               const zxc_decompress_opts_t* opts) {
    if (!buffer || comp_size < ZXC_FILE_HEADER_SIZE + ZXC_FILE_FOOTER_SIZE || comp_size > buffer_capacity)
        return ZXC_ERROR_NULL_INPUT;
    uint8_t* buf = (uint8_t*)buffer;
    const uint8_t* comp = buf + (buffer_capacity - comp_size);
    uint64_t dsize = 0, margin = 0, fl = 0;
    const int rc = inplace_probe(comp, comp_size, &dsize, &margin, &fl);
    if (rc != ZXC_OK) return rc;
    if (dsize > (uint64_t)buffer_capacity || (uint64_t)buffer_capacity - dsize < margin) return ZXC_ERROR_DST_TOO_SMALL;
    if ((uint64_t)(buffer_capacity - comp_size) < fl) return ZXC_ERROR_DST_TOO_SMALL;
    if (dsize == 0) return 0;
    return zxc_decompress(comp, comp_size, buf, buffer_capacity, opts);
}

/* ------------------------------------------------------------ zxc_compress */
static void wr64(uint8_t* p, uint64_t v) {
    wr32(p, (uint32_t)v);
    wr32(p + 4, (uint32_t)(v >> 32));
}

/* zxc_compress over batches of blocks (round 4): a helper thread uploads batch i + 1 and launches its encode on that slot's
 * stream while this thread turns batch i's block sizes into offsets, compacts the slots (zxc_gather_blocks_kernel) and brings
 * the bytes back — H2D, kernel and D2H of neighbouring batches overlap. One-shot this was 1 GiB in 56 ms = H2D + 29 ms of
 * kernel + D2H in series; batched 44 ms, now bound by the encode launches themselves (2 048 blocks = one workgroup per slot of
 * the chip, under the copies' traffic); a second upload thread and 96-384 MiB batches measured the same
 * (profiles/r4c_host_api.log; ZXC_MI355X_DEBUG_TIMES=1 prints the phases). */
typedef struct {
    void *d_src, *d_slots, *d_sizes, *d_offs, *d_out, *stream;
} comp_slot_t;
typedef struct {
    const uint8_t* src;
    size_t bytes;
    uint32_t block_size;
    int level, checksum, device, rc;
    comp_slot_t* s;
} comp_up_t;
static void* comp_up_main(void* arg) {
    comp_up_t* u = (comp_up_t*)arg;
    u->rc = zxc_mi355x_set_device(u->device);
    if (u->rc == ZXC_OK) u->rc = zxc_hip_memcpy_h2d_async(u->s->d_src, u->src, u->bytes, u->s->stream);
    if (u->rc == ZXC_OK)
        u->rc = zxc_mi355x_encode_blocks_device(u->s->d_src, u->bytes, u->block_size, u->level, u->checksum, u->s->d_slots,
                                                (uint32_t*)u->s->d_sizes, u->s->stream);
    return NULL;
}
static void comp_slot_free(comp_slot_t* s) {
    zxc_mi355x_free(s->d_src);
    zxc_mi355x_free(s->d_slots);
    zxc_mi355x_free(s->d_sizes);
    zxc_mi355x_free(s->d_offs);
    zxc_mi355x_free(s->d_out);
    zxc_hip_stream_destroy(s->stream);
    memset(s, 0, sizeof *s);
}
/* -> ZXC_OK and *op_io advanced over the nb encoded blocks, sizes[] and *hash_io filled; or a negative zxc_error_t */
static int compress_batches(const uint8_t* src, size_t src_size, size_t block_size, int level, int checksum_enabled, size_t tail_need,
                            uint8_t* dst, size_t dst_capacity, size_t* op_io, uint32_t* sizes, uint32_t nb, uint32_t batch_blocks,
                            uint32_t* hash_io) {
    const uint32_t stride = zxc_mi355x_encode_slot_stride((uint32_t)block_size);
    const size_t batch_bytes = (size_t)batch_blocks * block_size;
    const int device = zxc_hip_current_device();
    const int dbg = getenv("ZXC_MI355X_DEBUG_TIMES") != NULL;
    struct timespec t_[8];
#define TS(k) do { if (dbg) clock_gettime(CLOCK_MONOTONIC, &t_[k]); } while (0)
#define MS(a, b) ((t_[b].tv_sec - t_[a].tv_sec) * 1e3 + (t_[b].tv_nsec - t_[a].tv_nsec) / 1e6)
    TS(0);
    comp_slot_t S[2];
    memset(S, 0, sizeof S);
    uint64_t* offs = (uint64_t*)malloc((size_t)batch_blocks * sizeof(uint64_t));
    int rc = offs ? ZXC_OK : ZXC_ERROR_MEMORY;
    for (int k = 0; k < 2 && rc == ZXC_OK; k++) {
        S[k].d_src = zxc_mi355x_malloc(batch_bytes + 64);
        S[k].d_slots = zxc_mi355x_malloc((size_t)batch_blocks * stride);
        S[k].d_sizes = zxc_mi355x_malloc((size_t)batch_blocks * 4);
        S[k].d_offs = zxc_mi355x_malloc((size_t)batch_blocks * 8);
        S[k].d_out = zxc_mi355x_malloc((size_t)batch_blocks * (block_size + 64) + 64); /* (a block is never larger than RAW: header + bytes + trailer) */
        if (!S[k].d_src || !S[k].d_slots || !S[k].d_sizes || !S[k].d_offs || !S[k].d_out) rc = ZXC_ERROR_MEMORY;
        if (rc == ZXC_OK) rc = zxc_hip_stream_create(&S[k].stream);
    }
    size_t op = *op_io;
    uint32_t global_hash = *hash_io;
    const uint32_t nbatches = (nb + batch_blocks - 1) / batch_blocks;
    comp_up_t up;
    memset(&up, 0, sizeof up);
    TS(1);
    if (dbg) fprintf(stderr, "[zxc_compress] device buffers + streams %.2f ms\n", MS(0, 1));
    if (rc == ZXC_OK) { /* batch 0: uploaded and launched here */
        up = (comp_up_t){src, src_size < batch_bytes ? src_size : batch_bytes, (uint32_t)block_size, level, checksum_enabled, device, ZXC_OK, &S[0]};
        comp_up_main(&up);
        rc = up.rc;
    }
    for (uint32_t bi = 0; bi < nbatches && rc == ZXC_OK; bi++) {
        comp_slot_t* s = &S[bi & 1];
        const uint32_t b0 = bi * batch_blocks, nbi = nb - b0 < batch_blocks ? nb - b0 : batch_blocks;
        pthread_t th;
        int live = 0, started = 0;
        if (bi + 1 < nbatches) { /* the next batch, on the other slot (its last user, batch bi - 1, is done) */
            const size_t o = (size_t)(bi + 1) * batch_bytes;
            up = (comp_up_t){src + o, src_size - o < batch_bytes ? src_size - o : batch_bytes, (uint32_t)block_size, level, checksum_enabled, device, ZXC_OK, &S[(bi + 1) & 1]};
            started = 1;
            live = pthread_create(&th, NULL, comp_up_main, &up) == 0;
            if (!live) comp_up_main(&up); /* (no thread: in series) */
        }
        TS(2);
        rc = zxc_mi355x_synchronize(s->stream);
        TS(3);
        /* From here on this thread works on the default stream with synchronous calls: the runtime maps streams onto a few
         * hardware queues, and when this slot's stream shares one with the other slot's, anything queued on it now would
         * stand behind batch i + 1's upload and encode (measured inside a process that holds other streams: the 8 KiB of
         * sizes took 4.7 ms, the call ran at one-shot speed). */
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(sizes + b0, s->d_sizes, (size_t)nbi * 4);
        uint64_t total = 0;
        if (rc == ZXC_OK) {
            for (uint32_t i = 0; i < nbi; i++) {
                /* (never trusted as a copy length, nor as the place of the trailer the global hash is folded from: a block is at
                 *  least its 8-byte header, + 4 with checksums) */
                if (sizes[b0 + i] > block_size + 64 || sizes[b0 + i] < 8u + (checksum_enabled ? 4u : 0u)) { rc = ZXC_ERROR_CORRUPT_DATA; break; }
                offs[i] = total;
                total += sizes[b0 + i];
            }
            if (rc == ZXC_OK && (uint64_t)op + total + tail_need > (uint64_t)dst_capacity) rc = ZXC_ERROR_DST_TOO_SMALL;
        }
        TS(4);
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_h2d(s->d_offs, offs, (size_t)nbi * 8);
        if (rc == ZXC_OK)
            rc = zxc_mi355x_gather_blocks_device(s->d_slots, (uint32_t)block_size, (const uint32_t*)s->d_sizes, (const uint64_t*)s->d_offs,
                                                 s->d_out, nbi, NULL);
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(dst + op, s->d_out, (size_t)total); /* (ordered behind the gather, returns when the bytes are here) */
        if (rc == ZXC_OK && checksum_enabled) /* fold the block trailers in stream order (zxc_dispatch.c:754-759) */
            for (uint32_t i = 0; i < nbi; i++)
                global_hash = ((global_hash << 1) | (global_hash >> 31)) ^ rd32(dst + op + offs[i] + sizes[b0 + i] - 4);
        if (rc == ZXC_OK) op += (size_t)total;
        TS(5);
        if (live) pthread_join(th, NULL);
        TS(6);
        if (dbg) fprintf(stderr, "[zxc_compress] batch %u: wait for its encode %.2f, sizes %.2f, offsets + gather + download %.2f, wait for the next upload %.2f ms\n", bi, MS(2, 3), MS(3, 4), MS(4, 5), MS(5, 6));
        if (started && rc == ZXC_OK) rc = up.rc;
    }
    /* nothing of ours may still be running on the slots when they are freed */
    for (int k = 0; k < 2; k++)
        if (S[k].stream) (void)zxc_mi355x_synchronize(S[k].stream);
    TS(6);
    for (int k = 0; k < 2; k++) comp_slot_free(&S[k]);
    free(offs);
    TS(7);
    if (dbg) fprintf(stderr, "[zxc_compress] release %.2f ms\n", MS(6, 7));
#undef TS
#undef MS
    if (rc != ZXC_OK) return rc;
    *op_io = op;
    *hash_io = global_hash;
    return ZXC_OK;
}

int64_t zxc_compress(const void* src, const size_t src_size, void* dst_v, const size_t dst_capacity,
                     const zxc_compress_opts_t* opts) {
    uint8_t* dst = (uint8_t*)dst_v;
    if (!dst || dst_capacity == 0 || (src_size > 0 && !src)) return ZXC_ERROR_NULL_INPUT;
    const int checksum_enabled = opts ? opts->checksum_enabled : 0;
    const int seekable = opts ? opts->seekable : 0;
    int level = (opts && opts->level > 0) ? opts->level : ZXC_LEVEL_DEFAULT;
    if (level > ZXC_LEVEL_ULTRA) level = ZXC_LEVEL_ULTRA;
    const size_t block_size = (opts && opts->block_size > 0) ? opts->block_size : ZXC_BLOCK_SIZE_DEFAULT;
    const size_t dict_size = (opts && opts->dict) ? opts->dict_size : 0;
    if (dict_size > ZXC_DICT_SIZE_MAX) return ZXC_ERROR_DICT_TOO_LARGE;
    if (block_size < ZXC_BLOCK_SIZE_MIN || block_size > ZXC_BLOCK_SIZE_MAX || (block_size & (block_size - 1)))
        return ZXC_ERROR_BAD_BLOCK_SIZE;
    const uint8_t* dict = dict_size ? (const uint8_t*)opts->dict : NULL;
    const uint8_t* dict_huf = dict_size ? (const uint8_t*)opts->dict_huf : NULL;
    if (dst_capacity < ZXC_FILE_HEADER_SIZE) return ZXC_ERROR_DST_TOO_SMALL;

    /* file header (src/lib/zxc_common.c:534-558) */
    memset(dst, 0, ZXC_FILE_HEADER_SIZE);
    wr32(dst, MAGIC);
    dst[4] = FORMAT_VERSION;
    uint8_t lg = 0;
    while (((size_t)1 << lg) < block_size) lg++;
    dst[5] = lg;
    dst[6] = checksum_enabled ? 0x80 : 0; /* HAS_CHECKSUM | algo 0 (rapidhash) */
    if (dict_size) { /* HAS_DICTIONARY + dict_id binding content (and shared table), src/lib/zxc_common.c:546-553 */
        dst[6] |= 0x40;
        wr32(dst + 7, dict_id_of(dict, dict_size, dict_huf));
    }
    const uint16_t crc = hdr_hash16(dst);
    dst[14] = (uint8_t)crc;
    dst[15] = (uint8_t)(crc >> 8);
    size_t op = ZXC_FILE_HEADER_SIZE;

    const uint64_t nb64 = ((uint64_t)src_size + block_size - 1) / block_size;
    if (nb64 > 0x7FFFFFFFull) return ZXC_ERROR_BAD_BLOCK_SIZE;
    const uint32_t nb = (uint32_t)nb64;
    uint32_t* sizes = NULL;
    uint32_t global_hash = 0;
    size_t batch_bytes = FRAME_BATCH_BYTES;
    { const char* e = getenv("ZXC_MI355X_FRAME_BATCH_MIB"); if (e && atoi(e) >= 1 && atoi(e) <= 1024) batch_bytes = (size_t)atoi(e) << 20; }
    const uint32_t batch_blocks = (uint32_t)(batch_bytes / block_size > 0 ? batch_bytes / block_size : 1);
    if (nb > batch_blocks && !dict_size) { /* two batches or more: pipelined */
        if (zxc_mi355x_device_count() <= 0) return ZXC_ERROR_GPU_UNAVAILABLE;
        sizes = (uint32_t*)malloc((size_t)nb * sizeof(uint32_t));
        if (!sizes) return ZXC_ERROR_MEMORY;
        const size_t tail_need = BLK_HDR + (seekable ? zxc_seek_table_size(nb) : 0) + ZXC_FILE_FOOTER_SIZE;
        const int rc = compress_batches((const uint8_t*)src, src_size, block_size, level, checksum_enabled, tail_need, dst, dst_capacity,
                                        &op, sizes, nb, batch_blocks, &global_hash);
        if (rc != ZXC_OK) { free(sizes); return rc; }
    } else if (nb > 0) {
        if (zxc_mi355x_device_count() <= 0) return ZXC_ERROR_GPU_UNAVAILABLE;
        const uint32_t stride = zxc_mi355x_encode_slot_stride((uint32_t)block_size);
        sizes = (uint32_t*)malloc((size_t)nb * sizeof(uint32_t));
        uint64_t* offs = (uint64_t*)malloc((size_t)nb * sizeof(uint64_t));
        void* d_src = zxc_mi355x_malloc(src_size + 64);
        void* d_slots = zxc_mi355x_malloc((size_t)nb * stride);
        void* d_sizes = zxc_mi355x_malloc((size_t)nb * 4);
        void* d_offs = zxc_mi355x_malloc((size_t)nb * 8);
        void* d_out = NULL;
        void* d_dict = dict_size ? zxc_mi355x_malloc(dict_size + 64) : NULL;
        void* d_work = dict_size ? zxc_mi355x_malloc((size_t)zxc_mi355x_encode_dict_work_size(src_size, (uint32_t)block_size, (uint32_t)dict_size)) : NULL;
        int64_t rc = ZXC_ERROR_MEMORY;
        if (sizes && offs && d_src && d_slots && d_sizes && d_offs && (!dict_size || (d_dict && d_work))) {
            rc = zxc_mi355x_memcpy_h2d(d_src, src, src_size);
            if (rc == ZXC_OK && dict_size) rc = zxc_mi355x_memcpy_h2d(d_dict, dict, dict_size);
            if (rc == ZXC_OK)
                rc = dict_size ? zxc_mi355x_encode_blocks_dict_device(d_src, src_size, (uint32_t)block_size, level, checksum_enabled,
                                                                      d_dict, (uint32_t)dict_size, d_work, d_slots, (uint32_t*)d_sizes, NULL)
                               : zxc_mi355x_encode_blocks_device(d_src, src_size, (uint32_t)block_size, level, checksum_enabled,
                                                                 d_slots, (uint32_t*)d_sizes, NULL);
            if (rc == ZXC_OK) rc = zxc_mi355x_synchronize(NULL);
            if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(sizes, d_sizes, (size_t)nb * 4);
            uint64_t total = 0;
            if (rc == ZXC_OK) {
                for (uint32_t i = 0; i < nb; i++) { offs[i] = total; total += sizes[i]; }
                const uint64_t need = op + total + BLK_HDR + (seekable ? zxc_seek_table_size(nb) : 0) + ZXC_FILE_FOOTER_SIZE;
                if (need > dst_capacity) rc = ZXC_ERROR_DST_TOO_SMALL;
            }
            if (rc == ZXC_OK) {
                d_out = zxc_mi355x_malloc((size_t)total + 64);
                if (!d_out) rc = ZXC_ERROR_MEMORY;
            }
            if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_h2d(d_offs, offs, (size_t)nb * 8);
            if (rc == ZXC_OK)
                rc = zxc_mi355x_gather_blocks_device(d_slots, (uint32_t)block_size, (const uint32_t*)d_sizes,
                                                     (const uint64_t*)d_offs, d_out, nb, NULL);
            if (rc == ZXC_OK) rc = zxc_mi355x_synchronize(NULL);
            if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(dst + op, d_out, (size_t)total);
            if (rc == ZXC_OK && checksum_enabled) /* fold the block trailers in stream order (zxc_dispatch.c:754-759) */
                for (uint32_t i = 0; i < nb; i++)
                    global_hash = ((global_hash << 1) | (global_hash >> 31)) ^ rd32(dst + op + offs[i] + sizes[i] - 4);
            if (rc == ZXC_OK) op += (size_t)total;
        }
        zxc_mi355x_free(d_src);
        zxc_mi355x_free(d_slots);
        zxc_mi355x_free(d_sizes);
        zxc_mi355x_free(d_offs);
        zxc_mi355x_free(d_out);
        zxc_mi355x_free(d_dict);
        zxc_mi355x_free(d_work);
        free(offs);
        if (rc != ZXC_OK) { free(sizes); return rc; }
    }
    /* EOF block, optional seek table, footer (src/lib/zxc_dispatch.c:784-815) */
    if (dst_capacity - op < BLK_HDR) { free(sizes); return ZXC_ERROR_DST_TOO_SMALL; }
    memset(dst + op, 0, BLK_HDR);
    dst[op] = BLK_EOF;
    dst[op + 7] = hdr_hash8(dst + op);
    op += BLK_HDR;
    if (seekable && nb > 0) {
        const int64_t st = zxc_write_seek_table(dst + op, dst_capacity - op, sizes, nb);
        if (st < 0) { free(sizes); return st; }
        op += (size_t)st;
    }
    free(sizes);
    if (dst_capacity - op < ZXC_FILE_FOOTER_SIZE) return ZXC_ERROR_DST_TOO_SMALL;
    wr64(dst + op, (uint64_t)src_size);
    wr32(dst + op + 8, checksum_enabled ? global_hash : 0); /* zero when checksums are off */
    op += ZXC_FILE_FOOTER_SIZE;
    return (int64_t)op;
}

/* ---------------------------------------------------------------- seekable */
struct zxc_seekable_s {
    const uint8_t* src; /* borrowed; NULL in reader mode */
    uint64_t src_size;
    zxc_reader_t reader;
    uint32_t num_blocks;
    uint32_t block_size;
    uint64_t total_decomp;
    int file_has_checksums;
    uint32_t dict_id;
    uint32_t* comp_sizes;
    uint64_t* comp_offsets; /* [num_blocks + 1] */
    uint8_t* dict;          /* owned copy (zxc_seekable_set_dict) */
    size_t dict_size;
    uint8_t dict_huf[ZXC_HUF_TABLE_SIZE];
    int has_dict_huf;
};

size_t zxc_seek_table_size(const uint32_t num_blocks) { return BLK_HDR + (size_t)num_blocks * 4; }

int64_t zxc_write_seek_table(uint8_t* dst, const size_t dst_capacity, const uint32_t* comp_sizes,
                             const uint32_t num_blocks) {
    if (num_blocks > UINT32_MAX / 4) return ZXC_ERROR_OVERFLOW;
    const size_t total = zxc_seek_table_size(num_blocks);
    if (dst_capacity < total) return ZXC_ERROR_DST_TOO_SMALL;
    if (!dst || !comp_sizes) return ZXC_ERROR_NULL_INPUT;
    dst[0] = BLK_SEK;
    dst[1] = dst[2] = 0;
    wr32(dst + 3, num_blocks * 4);
    dst[7] = 0;
    dst[7] = hdr_hash8(dst);
    for (uint32_t i = 0; i < num_blocks; i++) wr32(dst + BLK_HDR + 4 * (size_t)i, comp_sizes[i]);
    return (int64_t)total;
}

void zxc_seekable_free(zxc_seekable* s) {
    if (!s) return;
    free(s->comp_sizes);
    free(s->comp_offsets);
    free(s->dict);
    free(s);
}

/* src/lib/zxc_seekable.c:1144-1174 */
int zxc_seekable_set_dict(zxc_seekable* s, const void* dict, size_t dict_size, const void* dict_huf) {
    if (!s || !dict || dict_size == 0) return ZXC_ERROR_NULL_INPUT;
    if (dict_size > ZXC_DICT_SIZE_MAX) return ZXC_ERROR_DICT_TOO_LARGE;
    if (s->dict_id != 0 && dict_id_of((const uint8_t*)dict, dict_size, (const uint8_t*)dict_huf) != s->dict_id)
        return ZXC_ERROR_DICT_MISMATCH;
    free(s->dict);
    s->dict = (uint8_t*)malloc(dict_size);
    s->dict_size = 0;
    s->has_dict_huf = 0;
    if (!s->dict) return ZXC_ERROR_MEMORY;
    memcpy(s->dict, dict, dict_size);
    s->dict_size = dict_size;
    if (dict_huf) {
        memcpy(s->dict_huf, dict_huf, ZXC_HUF_TABLE_SIZE);
        s->has_dict_huf = 1;
    }
    return ZXC_OK;
}

/* shared by both openers: entries = the SEK payload (num_blocks LE u32) */
static zxc_seekable* seekable_build(uint32_t block_size, int has_ck, uint32_t dict_id, uint64_t total,
                                    uint32_t nb, const uint8_t* entries, uint64_t archive_size) {
    zxc_seekable* s = (zxc_seekable*)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->num_blocks = nb;
    s->block_size = block_size;
    s->file_has_checksums = has_ck;
    s->dict_id = dict_id;
    s->total_decomp = total;
    s->src_size = archive_size;
    s->comp_sizes = (uint32_t*)calloc(nb, sizeof(uint32_t));
    s->comp_offsets = (uint64_t*)calloc((size_t)nb + 1, sizeof(uint64_t));
    if (!s->comp_sizes || !s->comp_offsets) { zxc_seekable_free(s); return NULL; }
    uint64_t acc = ZXC_FILE_HEADER_SIZE;
    for (uint32_t i = 0; i < nb; i++) {
        const uint32_t cs = rd32(entries + 4 * (size_t)i);
        if (cs < BLK_HDR || cs > archive_size) { zxc_seekable_free(s); return NULL; }
        s->comp_sizes[i] = cs;
        s->comp_offsets[i] = acc;
        acc += cs;
        if (acc > archive_size) { zxc_seekable_free(s); return NULL; }
    }
    s->comp_offsets[nb] = acc;
    return s;
}

zxc_seekable* zxc_seekable_open(const void* src_v, const size_t n) {
    const uint8_t* data = (const uint8_t*)src_v;
    if (!data || n < ZXC_FILE_HEADER_SIZE + 2 * BLK_HDR + ZXC_FILE_FOOTER_SIZE) return NULL;
    uint32_t bs, did;
    int ck;
    if (read_file_header(data, n, &bs, &ck, &did) != ZXC_OK) return NULL;
    const uint64_t total = rd64(data + n - ZXC_FILE_FOOTER_SIZE);
    if (total == 0) return NULL;
    const uint64_t nb = (total + bs - 1) / bs;
    if (nb > UINT32_MAX) return NULL;
    const uint64_t entries = nb * 4;
    if (entries + BLK_HDR + ZXC_FILE_FOOTER_SIZE > n) return NULL;
    const uint8_t* sek = data + n - ZXC_FILE_FOOTER_SIZE - BLK_HDR - (size_t)entries;
    uint8_t type;
    uint32_t csz;
    if (read_block_header(sek, BLK_HDR + (size_t)entries, &type, &csz) != ZXC_OK) return NULL;
    if (type != BLK_SEK || csz != (uint32_t)entries) return NULL;
    zxc_seekable* s = seekable_build(bs, ck, did, total, (uint32_t)nb, sek + BLK_HDR, n);
    if (!s) return NULL;
    s->src = data;
    /* the prefix sum must land on a real EOF block right in front of the SEK block */
    const uint64_t acc = s->comp_offsets[nb];
    if (acc != (uint64_t)(sek - data) - BLK_HDR || read_block_header(data + acc, BLK_HDR, &type, &csz) != ZXC_OK ||
        type != BLK_EOF) {
        zxc_seekable_free(s);
        return NULL;
    }
    return s;
}

zxc_seekable* zxc_seekable_open_reader(const zxc_reader_t* r) {
    if (!r || !r->read_at || r->size == 0) return NULL;
    if (r->size < ZXC_FILE_HEADER_SIZE + 2 * BLK_HDR + ZXC_FILE_FOOTER_SIZE) return NULL;
    uint8_t hdr[ZXC_FILE_HEADER_SIZE], foot[ZXC_FILE_FOOTER_SIZE];
    if (r->read_at(r->ctx, hdr, sizeof hdr, 0) != (int64_t)sizeof hdr) return NULL;
    uint32_t bs, did;
    int ck;
    if (read_file_header(hdr, sizeof hdr, &bs, &ck, &did) != ZXC_OK) return NULL;
    if (r->read_at(r->ctx, foot, sizeof foot, r->size - sizeof foot) != (int64_t)sizeof foot) return NULL;
    const uint64_t total = rd64(foot);
    if (total == 0) return NULL;
    const uint64_t nb = (total + bs - 1) / bs;
    if (nb > UINT32_MAX) return NULL;
    const uint64_t entries = nb * 4;
    if (entries + BLK_HDR + ZXC_FILE_FOOTER_SIZE > r->size) return NULL;
    const size_t sek_total = BLK_HDR + (size_t)entries;
    uint8_t* buf = (uint8_t*)malloc(sek_total);
    if (!buf) return NULL;
    zxc_seekable* s = NULL;
    uint8_t type;
    uint32_t csz;
    if (r->read_at(r->ctx, buf, sek_total, r->size - ZXC_FILE_FOOTER_SIZE - sek_total) == (int64_t)sek_total &&
        read_block_header(buf, sek_total, &type, &csz) == ZXC_OK && type == BLK_SEK && csz == (uint32_t)entries) {
        s = seekable_build(bs, ck, did, total, (uint32_t)nb, buf + BLK_HDR, r->size);
        if (s) s->reader = *r;
    }
    free(buf);
    return s;
}

uint32_t zxc_seekable_get_num_blocks(const zxc_seekable* s) { return s ? s->num_blocks : 0; }
uint64_t zxc_seekable_get_decompressed_size(const zxc_seekable* s) { return s ? s->total_decomp : 0; }
uint32_t zxc_seekable_get_block_comp_size(const zxc_seekable* s, const uint32_t i) {
    return (s && i < s->num_blocks) ? s->comp_sizes[i] : 0;
}
uint32_t zxc_seekable_get_block_decomp_size(const zxc_seekable* s, const uint32_t i) {
    if (!s || i >= s->num_blocks) return 0;
    const uint64_t rem = s->total_decomp - (uint64_t)i * s->block_size;
    return rem >= s->block_size ? s->block_size : (uint32_t)rem;
}

int64_t zxc_mi355x_plan_seekable(const zxc_seekable* s, uint32_t first, uint32_t n, uint64_t comp_rebase,
                                 zxc_dev_job_t* jobs) {
    if (!s || !jobs) return ZXC_ERROR_NULL_INPUT;
    if ((uint64_t)first + n > s->num_blocks) return ZXC_ERROR_SRC_TOO_SMALL;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t i = first + k;
        if (s->comp_offsets[i] < comp_rebase) return ZXC_ERROR_CORRUPT_DATA;
        jobs[k].comp_off = s->comp_offsets[i] - comp_rebase;
        jobs[k].comp_size = s->comp_sizes[i];
        jobs[k].out_off = (uint64_t)k * s->block_size;
        jobs[k].out_len = zxc_seekable_get_block_decomp_size(s, i);
    }
    return (int64_t)n;
}

/* Bytes [offset, offset + len) of the archive into dst, on the calling thread's device, arena `sub`, stream `stream`:
 * the covered blocks go to the device in batches of at most HOST_BATCH_BYTES of output slots (device memory is
 * O(batch) whatever the range), each batch = one launch over its blocks' compressed span. */
static int64_t seek_range_on(zxc_seekable* s, uint8_t* dst, const uint64_t offset, const size_t len, int sub, void* stream) {
    const dict_ref_t dr = {s->dict, s->dict_size, s->has_dict_huf ? s->dict_huf : NULL};
    const uint32_t b0 = (uint32_t)(offset / s->block_size);
    const uint32_t b1 = (uint32_t)((offset + len - 1) / s->block_size);
    uint32_t batch_blocks = (uint32_t)(HOST_BATCH_BYTES / s->block_size);
    if (batch_blocks < 16u) batch_blocks = 16u;
    const uint32_t nmax = (b1 - b0 + 1u) < batch_blocks ? (b1 - b0 + 1u) : batch_blocks;
    zxc_dev_job_t* jobs = (zxc_dev_job_t*)malloc((size_t)nmax * sizeof(*jobs));
    int32_t* st = (int32_t*)malloc((size_t)nmax * sizeof(int32_t));
    uint8_t* staged = NULL;
    size_t staged_cap = 0;
    int64_t ret = (jobs && st) ? (int64_t)len : (int64_t)ZXC_ERROR_MEMORY;
    for (uint32_t f = b0; ret >= 0 && f <= b1; f += nmax) {
        const uint32_t n = (b1 - f + 1u) < nmax ? (b1 - f + 1u) : nmax;
        const uint64_t c0 = s->comp_offsets[f], c1 = s->comp_offsets[f + n];
        if (c1 > s->src_size) { ret = ZXC_ERROR_SRC_TOO_SMALL; break; }
        const size_t comp_bytes = (size_t)(c1 - c0);
        /* compressed span of the batch's blocks: borrowed buffer or reader callback */
        const uint8_t* h_comp;
        if (s->src) {
            h_comp = s->src + c0;
        } else {
            if (staged_cap < comp_bytes) {
                free(staged);
                staged = (uint8_t*)malloc(comp_bytes ? comp_bytes : 1);
                staged_cap = staged ? comp_bytes : 0;
                if (!staged) { ret = ZXC_ERROR_MEMORY; break; }
            }
            const int64_t r = s->reader.read_at(s->reader.ctx, staged, comp_bytes, c0);
            if (r != (int64_t)comp_bytes) { ret = r < 0 ? r : (int64_t)ZXC_ERROR_IO; break; }
            h_comp = staged;
        }
        zxc_mi355x_plan_seekable(s, f, n, c0, jobs);
        /* The reference decodes each block with cap block_size + 2112 and keeps what
         * the range needs (zxc_seekable.c:758-780): keep whole slots here. */
        for (uint32_t k = 0; k < n; k++) jobs[k].out_len = s->block_size;
        dev_bufs_t b;
        const int rc = run_jobs_on(h_comp, comp_bytes, jobs, n, (size_t)n * s->block_size, s->block_size, 0u, 0, st, &b, &dr, sub, stream);
        if (rc != ZXC_OK) { ret = rc; break; }
        /* first failing block in job order wins (zxc_seekable.c:1097-1104) */
        const uint64_t lo = (uint64_t)f * s->block_size;                 /* decoded position of the batch's first byte */
        const uint64_t from = offset > lo ? offset : lo;
        const uint64_t batch_end = lo + (uint64_t)n * s->block_size;
        const uint64_t to = (offset + len) < batch_end ? (offset + len) : batch_end;
        for (uint32_t k = 0; k < n; k++) {
            if (st[k] < 0) { ret = st[k]; break; }
            /* a block that decodes short of what the range needs from it */
            const uint64_t bstart = lo + (uint64_t)k * s->block_size;
            const uint64_t bneed_hi = to < bstart + zxc_seekable_get_block_decomp_size(s, f + k) ? to : bstart + zxc_seekable_get_block_decomp_size(s, f + k);
            if (bneed_hi > bstart && (uint64_t)st[k] < bneed_hi - bstart) { ret = ZXC_ERROR_CORRUPT_DATA; break; }
        }
        if (ret >= 0 && to > from) {
            const int crc = dev_bufs_read(&b, dst + (size_t)(from - offset), (size_t)(from - lo), (size_t)(to - from));
            if (crc != ZXC_OK) ret = crc;
        }
        dev_bufs_free(&b);
    }
    free(jobs);
    free(st);
    free(staged);
    return ret;
}

static int64_t seek_range_check(zxc_seekable* s, void* dst, const size_t dst_capacity, const uint64_t offset, const size_t len) {
    if (!s || !dst) return ZXC_ERROR_NULL_INPUT;
    if (dst_capacity < len) return ZXC_ERROR_DST_TOO_SMALL;
    if (offset > s->total_decomp || len > s->total_decomp - offset) return ZXC_ERROR_SRC_TOO_SMALL; /* (no wrap) */
    if (s->dict_id != 0 && (!s->dict || s->dict_size == 0)) return ZXC_ERROR_DICT_REQUIRED;
    return 0;
}

int64_t zxc_seekable_decompress_range(zxc_seekable* s, void* dst, const size_t dst_capacity, const uint64_t offset,
                                      const size_t len) {
    if (len == 0) return 0;
    const int64_t chk = seek_range_check(s, dst, dst_capacity, offset, len);
    if (chk < 0) return chk;
    return seek_range_on(s, (uint8_t*)dst, offset, len, 0, NULL);
}

/* The multi-threaded range decode (reference: zxc_seekable.c:1033-1108 plans one job per block and lets n_threads workers
 * pull them). Here a block is a workgroup's work, so host threads add nothing on ONE device; what they are for is
 * SEVERAL devices, and only on request: with ZXC_MI355X_DEVICES set (a comma-separated list of ordinals; an ordinal may
 * repeat: two workers, two streams on that device) the covered blocks are cut into min(n_threads, listed devices)
 * contiguous parts (n_threads == 0: all listed), one host thread + stream + staging arena per part. Without the variable
 * the call stays on the calling thread's current device (zxc_mi355x_set_device): in a one-rank-per-GPU job a rank must
 * never touch the other ranks' GPUs. First failing part in block order wins, like the reference's job scan. */
typedef struct {
    zxc_seekable* s;
    uint8_t* dst;
    uint64_t offset;
    size_t len;
    int device, sub;
    int64_t result;
} seek_part_t;
static void* seek_part_main(void* p) {
    seek_part_t* a = (seek_part_t*)p;
    void* stream = NULL;
    if (zxc_mi355x_set_device(a->device) != ZXC_OK || zxc_hip_stream_create(&stream) != ZXC_OK) {
        a->result = ZXC_ERROR_GPU_UNAVAILABLE;
        return NULL;
    }
    a->result = seek_range_on(a->s, a->dst, a->offset, a->len, a->sub, stream);
    zxc_hip_stream_destroy(stream);
    return NULL;
}
static int device_list(int* devs, int cap) {
    int count = zxc_mi355x_device_count();
    if (count > HOST_MAX_DEVICES) count = HOST_MAX_DEVICES; /* (ordinals index per-device tables of that size) */
    int n = 0;
    const char* e = getenv("ZXC_MI355X_DEVICES");
    if (!e || !*e) return 0; /* no explicit opt-in: no fan-out (the caller's current device) */
    while (*e && n < cap) {
        char* end = NULL;
        const long v = strtol(e, &end, 10);
        if (end == e) break;
        if (v >= 0 && v < count) devs[n++] = (int)v;
        e = (*end == ',') ? end + 1 : end;
        if (*end != ',' && *end != 0) break;
    }
    return n;
}

int64_t zxc_seekable_decompress_range_mt(zxc_seekable* s, void* dst, const size_t dst_capacity, const uint64_t offset,
                                         const size_t len, int n_threads) {
    if (len == 0) return 0;
    const int64_t chk = seek_range_check(s, dst, dst_capacity, offset, len);
    if (chk < 0) return chk;
    int devs[HOST_MAX_DEVICES * ARENA_SUBS];
    int nd = device_list(devs, (int)(sizeof devs / sizeof devs[0]));
    const uint32_t b0 = (uint32_t)(offset / s->block_size);
    const uint32_t b1 = (uint32_t)((offset + len - 1) / s->block_size);
    const uint32_t nblocks = b1 - b0 + 1u;
    int parts = n_threads == 0 ? nd : (n_threads < nd ? n_threads : nd);
    if ((uint32_t)parts > nblocks) parts = (int)nblocks;
    if (parts <= 1) return zxc_seekable_decompress_range(s, dst, dst_capacity, offset, len);
    seek_part_t part[HOST_MAX_DEVICES * ARENA_SUBS];
    pthread_t th[HOST_MAX_DEVICES * ARENA_SUBS];
    int live[HOST_MAX_DEVICES * ARENA_SUBS];
    int used[HOST_MAX_DEVICES];
    memset(used, 0, sizeof used);
    for (int k = 0; k < parts; k++) {
        /* blocks [b0 + k*N/P, b0 + (k+1)*N/P): zxc_mi355x block_range's cut, in bytes of the range */
        const uint32_t f = b0 + (uint32_t)(((uint64_t)nblocks * (uint32_t)k) / (uint32_t)parts);
        const uint32_t l = b0 + (uint32_t)(((uint64_t)nblocks * (uint32_t)(k + 1)) / (uint32_t)parts);
        const uint64_t lo = (uint64_t)f * s->block_size, hi = (uint64_t)l * s->block_size;
        const uint64_t from = offset > lo ? offset : lo, to = (offset + len) < hi ? (offset + len) : hi;
        part[k].s = s;
        part[k].dst = (uint8_t*)dst + (size_t)(from - offset);
        part[k].offset = from;
        part[k].len = (size_t)(to - from);
        part[k].device = devs[k];
        part[k].sub = used[devs[k]]++ % ARENA_SUBS;
        part[k].result = ZXC_ERROR_GPU_UNAVAILABLE;
        live[k] = 0;
        if (part[k].len == 0) { part[k].result = 0; continue; }
        if (pthread_create(&th[k], NULL, seek_part_main, &part[k]) != 0) part[k].result = ZXC_ERROR_MEMORY;
        else live[k] = 1;
    }
    int64_t ret = (int64_t)len;
    for (int k = 0; k < parts; k++)
        if (live[k]) pthread_join(th[k], NULL);
    for (int k = 0; k < parts; k++)
        if (part[k].result < 0) { ret = part[k].result; break; }
    return ret;
}


/* ------------------------------------------------------------ Block API + contexts */
/* reference include/zxc_buffer.h:204-468; impl src/lib/zxc_dispatch.c:1234-1858, bounds src/lib/zxc_common.c:873-902.
 * One block per call = one workgroup launch: correct, not fast — callers with many blocks should use the
 * frame / seekable APIs (one launch for all blocks) or the device-resident entry points of zxc_mi355x.h.
 * The contexts only carry the sticky options: the working memory of this library lives on the device
 * (per-device arena above), not in the context. */
#define BLOCK_FORMAT_OVERHEAD 68u /* ZXC_BLOCK_FORMAT_OVERHEAD, src/lib/zxc_internal.h:427 */
struct zxc_cctx_s { int level; size_t block_size; int checksum; };
struct zxc_dctx_s { int unused; };

uint32_t zxc_get_dict_id(const void* src, const size_t src_size) {
    if (!src || src_size < ZXC_FILE_HEADER_SIZE) return 0;
    const uint8_t* p = (const uint8_t*)src;
    if (rd32(p) != MAGIC) return 0;
    return (p[6] & 0x40) ? rd32(p + 7) : 0;
}

uint64_t zxc_compress_block_bound(const size_t input_size) {
    if (input_size == 0 || input_size > ZXC_BLOCK_SIZE_MAX) return 0;
    return (uint64_t)BLK_HDR + (uint64_t)input_size + BLOCK_FORMAT_OVERHEAD + 4u;
}

uint64_t zxc_decompress_block_bound(const size_t uncompressed_size) {
    if (uncompressed_size > ZXC_BLOCK_SIZE_MAX) return 0;
    return (uint64_t)uncompressed_size + TAIL_PAD;
}

static size_t block_size_ceil(size_t v) { /* zxc_block_size_ceil, src/lib/zxc_internal.h:885-896 */
    size_t bs = ZXC_BLOCK_SIZE_MIN;
    while (bs < v) bs <<= 1;
    return bs;
}

/* reference include/zxc_buffer.h:382: estimated peak working memory of one zxc_compress_block call of src_size bytes. Here
 * the working memory is DEVICE memory of the staging arena: the source block, its output slot (2 x block + 512) and, at
 * levels 6-7, the four PivCo level buffers; 0 for src_size 0, sizes round up to the block-size tiers. */
uint64_t zxc_estimate_cctx_size(const size_t src_size, const int level) {
    if (src_size == 0) return 0;
    const uint64_t bs = block_size_ceil(src_size < ZXC_BLOCK_SIZE_MIN ? ZXC_BLOCK_SIZE_MIN : src_size);
    uint64_t est = sizeof(struct zxc_cctx_s) + bs + 64u + (2u * bs + 512u) + 4096u;
    if (level >= 6) est += 4u * (bs + 64u);
    return est;
}

zxc_cctx* zxc_create_cctx(const zxc_compress_opts_t* opts) {
    const size_t bs = (opts && opts->block_size > 0) ? opts->block_size : ZXC_BLOCK_SIZE_DEFAULT;
    if (bs < ZXC_BLOCK_SIZE_MIN || bs > ZXC_BLOCK_SIZE_MAX || (bs & (bs - 1))) return NULL;
    zxc_cctx* c = (zxc_cctx*)calloc(1, sizeof(*c));
    if (!c) return NULL;
    int level = (opts && opts->level > 0) ? opts->level : ZXC_LEVEL_DEFAULT;
    c->level = level > ZXC_LEVEL_ULTRA ? ZXC_LEVEL_ULTRA : level;
    c->block_size = bs;
    c->checksum = opts ? opts->checksum_enabled : 0;
    return c;
}
void zxc_free_cctx(zxc_cctx* cctx) { free(cctx); }
zxc_dctx* zxc_create_dctx(void) { return (zxc_dctx*)calloc(1, sizeof(zxc_dctx)); }
void zxc_free_dctx(zxc_dctx* dctx) { free(dctx); }

/* zxc_compress with sticky options (src/lib/zxc_dispatch.c:1330-1430) */
int64_t zxc_compress_cctx(zxc_cctx* cctx, const void* src, const size_t src_size, void* dst, const size_t dst_capacity,
                          const zxc_compress_opts_t* opts) {
    if (!cctx) return ZXC_ERROR_NULL_INPUT;
    zxc_compress_opts_t o;
    memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    if (o.level <= 0) o.level = cctx->level;
    if (o.block_size == 0) o.block_size = cctx->block_size;
    if (!opts) o.checksum_enabled = cctx->checksum;
    if (o.level > ZXC_LEVEL_ULTRA) o.level = ZXC_LEVEL_ULTRA;
    if (o.block_size < ZXC_BLOCK_SIZE_MIN || o.block_size > ZXC_BLOCK_SIZE_MAX || (o.block_size & (o.block_size - 1)))
        return ZXC_ERROR_BAD_BLOCK_SIZE;
    cctx->level = o.level;
    cctx->block_size = o.block_size;
    cctx->checksum = o.checksum_enabled;
    return zxc_compress(src, src_size, dst, dst_capacity, &o);
}

int64_t zxc_decompress_dctx(zxc_dctx* dctx, const void* src, const size_t src_size, void* dst, const size_t dst_capacity,
                            const zxc_decompress_opts_t* opts) {
    if (!dctx) return ZXC_ERROR_NULL_INPUT;
    return zxc_decompress(src, src_size, dst, dst_capacity, opts);
}

int64_t zxc_compress_block(zxc_cctx* cctx, const void* src, const size_t src_size, void* dst, const size_t dst_capacity,
                           const zxc_compress_opts_t* opts) {
    if (!cctx || !src || !dst || src_size == 0 || dst_capacity == 0) return ZXC_ERROR_NULL_INPUT;
    if (src_size > ZXC_BLOCK_SIZE_MAX) return ZXC_ERROR_BAD_BLOCK_SIZE;
    const int checksum_enabled = opts ? opts->checksum_enabled : cctx->checksum;
    int level = (opts && opts->level > 0) ? opts->level : cctx->level;
    if (level > ZXC_LEVEL_ULTRA) level = ZXC_LEVEL_ULTRA;
    const uint8_t* b_dict = (opts && opts->dict && opts->dict_size > 0) ? (const uint8_t*)opts->dict : NULL;
    const size_t b_dict_size = b_dict ? opts->dict_size : 0;
    if (b_dict_size > ZXC_DICT_SIZE_MAX) return ZXC_ERROR_DICT_TOO_LARGE;
    const size_t want_bs = (opts && opts->block_size > 0) ? opts->block_size : cctx->block_size;
    const size_t min_bs = block_size_ceil(src_size);
    const size_t bs = want_bs > min_bs ? want_bs : min_bs; /* one block: block_size >= src_size */
    if (bs > ZXC_BLOCK_SIZE_MAX || (bs & (bs - 1))) return ZXC_ERROR_BAD_BLOCK_SIZE;
    cctx->level = level;
    cctx->block_size = bs;
    cctx->checksum = checksum_enabled;
    if (zxc_mi355x_device_count() <= 0) return ZXC_ERROR_GPU_UNAVAILABLE;
    const uint32_t stride = zxc_mi355x_encode_slot_stride((uint32_t)bs);
    void* d_src = zxc_mi355x_malloc(src_size + 64);
    void* d_slot = zxc_mi355x_malloc(stride);
    void* d_size = zxc_mi355x_malloc(4);
    void* d_dict = b_dict_size ? zxc_mi355x_malloc(b_dict_size + 64) : NULL;
    void* d_work = b_dict_size ? zxc_mi355x_malloc((size_t)zxc_mi355x_encode_dict_work_size(src_size, (uint32_t)bs, (uint32_t)b_dict_size)) : NULL;
    int64_t rc = ZXC_ERROR_MEMORY;
    uint32_t csize = 0;
    if (d_src && d_slot && d_size && (!b_dict_size || (d_dict && d_work))) {
        rc = zxc_mi355x_memcpy_h2d(d_src, src, src_size);
        if (rc == ZXC_OK && b_dict_size) rc = zxc_mi355x_memcpy_h2d(d_dict, b_dict, b_dict_size);
        if (rc == ZXC_OK)
            rc = b_dict_size ? zxc_mi355x_encode_blocks_dict_device(d_src, src_size, (uint32_t)bs, level, checksum_enabled, d_dict,
                                                                    (uint32_t)b_dict_size, d_work, d_slot, (uint32_t*)d_size, NULL)
                             : zxc_mi355x_encode_blocks_device(d_src, src_size, (uint32_t)bs, level, checksum_enabled, d_slot,
                                                               (uint32_t*)d_size, NULL);
        if (rc == ZXC_OK) rc = zxc_mi355x_synchronize(NULL);
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(&csize, d_size, 4);
        if (rc == ZXC_OK && csize > dst_capacity) rc = ZXC_ERROR_DST_TOO_SMALL;
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(dst, d_slot, csize);
    }
    zxc_mi355x_free(d_src);
    zxc_mi355x_free(d_slot);
    zxc_mi355x_free(d_size);
    zxc_mi355x_free(d_dict);
    zxc_mi355x_free(d_work);
    return rc == ZXC_OK ? (int64_t)csize : rc;
}

/* One block -> dst. `cap_override` 0: decode capacity block_size_ceil(dst_capacity) + 2112 like the
 * reference's work_buf bounce (zxc_dispatch.c:1745-1790); else the strict capacity of the safe variant. */
static int64_t decompress_one_block(const void* src, size_t src_size, void* dst, size_t dst_capacity,
                                    const zxc_decompress_opts_t* opts, uint32_t cap_override) {
    const int verify = opts ? opts->checksum_enabled : 0;
    const uint8_t* dict = opts ? (const uint8_t*)opts->dict : NULL;
    const size_t dict_size = (opts && opts->dict) ? opts->dict_size : 0;
    if (dict_size > ZXC_DICT_SIZE_MAX) return ZXC_ERROR_DICT_TOO_LARGE;
    const dict_ref_t dr = {dict, dict_size, (opts && opts->dict) ? (const uint8_t*)opts->dict_huf : NULL};
    size_t bs = block_size_ceil(dst_capacity);
    if (bs > ZXC_BLOCK_SIZE_MAX) bs = ZXC_BLOCK_SIZE_MAX;
    const size_t work = cap_override ? cap_override : bs + TAIL_PAD;
    zxc_dev_job_t job;
    job.comp_off = 0;
    job.out_off = 0;
    job.comp_size = src_size > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)src_size;
    job.out_len = (uint32_t)(dst_capacity < work ? dst_capacity : work);
    int32_t st = 0;
    dev_bufs_t b;
    /* only the block itself is uploaded: header + payload (+ trailer), never more than the caller's buffer */
    size_t up = src_size;
    if (src_size >= BLK_HDR) {
        const uint64_t phys = (uint64_t)BLK_HDR + rd32((const uint8_t*)src + 3) + 4u;
        if (phys < up) up = (size_t)phys;
    }
    int rc = run_jobs_cap((const uint8_t*)src, up, &job, 1, ((size_t)job.out_len + 15u) & ~(size_t)15u, (uint32_t)bs, cap_override,
                          verify, &st, &b, &dr);
    if (rc != ZXC_OK) return rc;
    int64_t ret = st;
    if (st > 0) {
        if ((size_t)st > dst_capacity) ret = ZXC_ERROR_DST_TOO_SMALL;
        else {
            rc = zxc_mi355x_memcpy_d2h(dst, b.d_out, (size_t)st);
            if (rc != ZXC_OK) ret = rc;
        }
    }
    dev_bufs_free(&b);
    return ret;
}

int64_t zxc_decompress_block(zxc_dctx* dctx, const void* src, const size_t src_size, void* dst, const size_t dst_capacity,
                             const zxc_decompress_opts_t* opts) {
    if (!dctx || !src || !dst || src_size < BLK_HDR || dst_capacity == 0) return ZXC_ERROR_NULL_INPUT;
    if (dst_capacity > (size_t)ZXC_BLOCK_SIZE_MAX + TAIL_PAD) return ZXC_ERROR_BAD_BLOCK_SIZE;
    return decompress_one_block(src, src_size, dst, dst_capacity, opts, 0u);
}

int64_t zxc_decompress_block_safe(zxc_dctx* dctx, const void* src, const size_t src_size, void* dst,
                                  const size_t dst_capacity, const zxc_decompress_opts_t* opts) {
    if (!dctx || !src || !dst || src_size < BLK_HDR || dst_capacity == 0) return ZXC_ERROR_NULL_INPUT;
    if (dst_capacity > ZXC_BLOCK_SIZE_MAX) return ZXC_ERROR_BAD_BLOCK_SIZE;
    /* dictionary inputs and RAW blocks take the bounce-capable path (zxc_dispatch.c:1823-1832) */
    if ((opts && opts->dict && opts->dict_size > 0) || ((const uint8_t*)src)[0] == BLK_RAW)
        return zxc_decompress_block(dctx, src, src_size, dst, dst_capacity, opts);
    return decompress_one_block(src, src_size, dst, dst_capacity, opts, (uint32_t)dst_capacity);
}

#include "zxc_stream_host.inc"
