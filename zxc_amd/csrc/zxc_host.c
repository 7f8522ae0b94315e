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
int zxc_hip_event_create(void** ev_out);
void zxc_hip_event_destroy(void* ev);
int zxc_hip_event_record(void* ev, void* stream);
int zxc_hip_event_synchronize(void* ev);
int zxc_hip_block_offsets(uint32_t* d_sizes, uint64_t* d_offsets, uint32_t n_blocks, uint32_t max_size, void* stream);
void* zxc_hip_host_alloc(size_t bytes);
void zxc_hip_host_free(void* p);

/* Staging arenas: the device buffers of the host API live across calls (grown on demand), so a call costs its copies
 * and its launch, not four hipMalloc + hipFree (which synchronise the device). One call at a time uses an arena; its
 * mutex is held from run_jobs() until dev_bufs_free(). Every device has ARENA_SUBS of them: sub 0 serves the
 * single-threaded API, the workers of zxc_seekable_decompress_range_mt take one each (two workers on one device run on
 * two streams side by side). Every caller batches its work (HOST_BATCH_BYTES of output slots per launch), so an arena
 * stays within a few hundred MiB; a buffer that grew beyond ARENA_KEEP_MAX is given back when the call ends, and
 * zxc_mi355x_release_cached() frees everything that is not in use. */
typedef struct { void* p; size_t cap; } dbuf_t;
enum { AR_COMP = 0, AR_JOBS, AR_OUT, AR_STATUS, AR_DICT, AR_SLOTS, AR_N }; /* (AR_SLOTS: the encoder's per-block output slots) */
typedef struct {
    pthread_mutex_t mu;
    dbuf_t buf[AR_N];
    int32_t* h_st;   /* pinned host memory for a piece's block statuses (the piece pipeline below) */
    size_t h_st_cap;
    void* ev;        /* event recorded behind a piece's launch + status copy */
    void* stream;    /* the stream this arena's pieces run on (created once: creating and destroying streams costs a call 0.3-0.5 ms) */
} arena_t;
#define HOST_MAX_DEVICES 16
#define ARENA_SUBS 8
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
            any |= a->h_st != NULL || a->ev != NULL || a->stream != NULL;
            if (any && zxc_mi355x_set_device(dev) == ZXC_OK) {
                for (int w = 0; w < AR_N; w++) { zxc_mi355x_free(a->buf[w].p); a->buf[w].p = NULL; a->buf[w].cap = 0; }
                zxc_hip_host_free(a->h_st); a->h_st = NULL; a->h_st_cap = 0;
                zxc_hip_event_destroy(a->ev); a->ev = NULL;
                zxc_hip_stream_destroy(a->stream); a->stream = NULL;
            }
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

/* ------------------------------------------------------------ the piece pipeline */
/* zxc_decompress and the seekable range calls move their blocks through the device in PIECES, three stages in flight. Two
 * producer threads take turns asking the call's SOURCE for the next piece (a run of blocks: their compressed span on the host + a
 * job table); each uploads its piece and enqueues its launch and the copy of its statuses on its own stream — a producer never
 * waits for the device, and while one is uploading the other already walks and uploads the piece behind it (the runtime's pageable
 * upload path moves ~25-40 GB/s per thread on this box, half the link). The calling thread waits for piece i's event and hands it
 * to the call's SINK, which checks the statuses in stream order (first failing block wins, like the reference's sequential loops)
 * and copies what the caller wants of the piece's output back. PCIe carries both directions at once and the launches (0.3-0.6 ms
 * from enqueue to done whatever the piece's size: one round of wavefronts) hide under the copies: a call is bounded by its
 * download (round 4: upload + launch of piece i+1 were one helper-thread job with a join behind every piece, the range call strictly
 * serial: 45 / 33 GB/s of a link that carries 56 GB/s one way, 49 each way at once — profiles/r5a_*, r5d_*). Pieces ramp up
 * (4, 8, 16, 32 MiB of output) so that the first download starts early. Device memory: one staging arena per slot in flight (up to
 * PIPE_SLOTS, whatever is free: a caller never WAITS for a second arena while holding one; with one arena, or for a frame of one
 * piece, the same steps run in series on the calling thread). */
#define PIPE_SLOTS 6        /* the most a call holds (zxc_compress); the decode calls ask for PIPE_DECODE_SLOTS */
#define PIPE_DECODE_SLOTS 4
#define PIPE_PRODUCERS 2
#define PIPE_FIRST_BYTES ((size_t)4 << 20)
#define PIPE_PIECE_BYTES ((size_t)32 << 20)
#define PIPE_IRREGULAR 1 /* a sink's verdict: stop here, the caller takes this piece another way */

typedef struct {
    const uint8_t* h_comp;   /* host span of the piece's blocks */
    size_t comp_bytes;
    zxc_dev_job_t* jobs;     /* (the slot's array, max_blocks entries) comp_off relative to h_comp, out_off = the block's slot in the piece's output */
    uint32_t n;
    size_t out_bytes;
    uint64_t cookie[6];      /* source -> sink */
    uint8_t* stage;          /* host staging a source may use (reader callbacks), kept by the slot across pieces */
    size_t stage_cap;
} pipe_piece_t;
typedef int (*pipe_source_fn)(void* ctx, uint32_t max_blocks, pipe_piece_t* p); /* 1: a piece, 0: no more, < 0: error */
typedef int (*pipe_sink_fn)(void* ctx, const pipe_piece_t* p, const int32_t* st, const dev_bufs_t* b); /* (st: the piece's pinned status words) 0: go on, else: stop with it */

typedef struct {
    arena_t* a;
    dev_bufs_t b;
    pipe_piece_t p;
    int rc;
    int dict_up;       /* the dictionary is in this slot's arena */
    uint64_t ready;    /* i + 1 once piece i sits enqueued (or failed: rc) in this slot */
} pipe_slot_t;

struct pipe_s;
typedef int (*pipe_enqueue_fn)(struct pipe_s* e, pipe_slot_t* s, void* stream); /* a call's own upload + launch (NULL: the block decode below) */
typedef struct pipe_s {
    pipe_source_fn source;
    void* source_ctx;
    pipe_enqueue_fn enqueue;
    void* enqueue_ctx;
    uint32_t block_size;
    int verify;
    const dict_ref_t* dr;
    uint32_t first_blocks, max_blocks, cap_blocks;
    size_t cap_comp, cap_out; /* what a slot reserves up front (no buffer grows while pieces are in flight) */
    int device, k;
    int producers;                /* (piece i is uploaded and launched on the stream of its slot's arena) */
    pipe_slot_t slot[PIPE_SLOTS];
    pthread_mutex_t mu;
    pthread_cond_t cv;
    pthread_mutex_t src_mu;       /* the source is asked for one piece at a time, in order */
    uint64_t next, consumed;      /* pieces handed out by the source / taken by the sink */
    int src_done, stop;
} pipe_t;

static double pipe_now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec * 1e3 + (double)t.tv_nsec * 1e-6;
}
#define PIPE_DBG_MAX 64
typedef struct { double src0, up0, up1, launched, ready, sunk; uint32_t n; size_t comp, out; } pipe_dbg_t;
static pipe_dbg_t* g_pipe_dbg = NULL; /* ZXC_MI355X_DEBUG_TIMES=1: one caller at a time, a timeline of the pieces on stderr */

static uint32_t pipe_piece_blocks(const pipe_t* e, uint64_t i) { /* the ramp: 4, 8, 16, ... MiB of output up to the piece size */
    uint32_t want = e->first_blocks;
    for (uint64_t r = 0; r < i && want < e->max_blocks; r++) want <<= 1;
    return want > e->max_blocks ? e->max_blocks : want;
}

/* piece i, already described by the source in its slot: upload, enqueue launch + status copy + event on `stream`. ZXC_OK or an error */
static int pipe_enqueue(pipe_t* e, uint64_t i) {
    pipe_slot_t* s = &e->slot[i % (uint64_t)e->k];
    void* stream = s->a->stream;
    pipe_dbg_t* dbg = (g_pipe_dbg && i < PIPE_DBG_MAX) ? &g_pipe_dbg[i] : NULL;
    if (dbg) { dbg->up0 = pipe_now_ms(); dbg->n = s->p.n; dbg->comp = s->p.comp_bytes; dbg->out = s->p.out_bytes; }
    arena_t* a = s->a;
    const uint32_t n = s->p.n;
    dev_bufs_t* b = &s->b;
    memset(b, 0, sizeof *b);
    b->stream = stream;
    if (a->h_st_cap < (size_t)e->max_blocks * sizeof(int32_t)) {
        zxc_hip_host_free(a->h_st);
        a->h_st_cap = 0;
        a->h_st = (int32_t*)zxc_hip_host_alloc((size_t)e->max_blocks * sizeof(int32_t));
        if (a->h_st) a->h_st_cap = (size_t)e->max_blocks * sizeof(int32_t);
    }
    if (!a->ev && zxc_hip_event_create(&a->ev) != ZXC_OK) a->ev = NULL;
    if (!a->h_st) return ZXC_ERROR_MEMORY;
    if (!a->ev) return ZXC_ERROR_GPU_UNAVAILABLE;
    int rc;
    if (e->enqueue) {
        rc = e->enqueue(e, s, stream);
        if (dbg) dbg->up1 = pipe_now_ms();
    } else {
        const size_t need_comp = (s->p.comp_bytes > e->cap_comp ? s->p.comp_bytes : e->cap_comp) + 64; /* (+64: 16-byte reads past the last block) */
        const size_t need_out = (s->p.out_bytes > e->cap_out ? s->p.out_bytes : e->cap_out) + 64;
        b->d_comp = arena_reserve(a, AR_COMP, need_comp);
        b->d_jobs = arena_reserve(a, AR_JOBS, (size_t)e->max_blocks * sizeof(zxc_dev_job_t));
        b->d_out = arena_reserve(a, AR_OUT, need_out);
        b->d_status = arena_reserve(a, AR_STATUS, (size_t)e->max_blocks * sizeof(int32_t));
        if (!b->d_comp || !b->d_jobs || !b->d_out || !b->d_status) return ZXC_ERROR_MEMORY;
        rc = zxc_hip_memcpy_h2d_async(b->d_comp, s->p.h_comp, s->p.comp_bytes, stream);
        if (rc == ZXC_OK) rc = zxc_hip_memcpy_h2d_async(b->d_jobs, s->p.jobs, (size_t)n * sizeof(zxc_dev_job_t), stream);
        const dict_ref_t* dr = e->dr;
        if (rc == ZXC_OK && dr && dr->dict_size) {
            b->d_dict = arena_reserve(a, AR_DICT, dr->dict_size + ZXC_HUF_TABLE_SIZE + 64);
            if (!b->d_dict) rc = ZXC_ERROR_MEMORY;
            if (rc == ZXC_OK && !s->dict_up) { /* once per slot and call (plain copies: they return when the bytes are there) */
                rc = zxc_mi355x_memcpy_h2d(b->d_dict, dr->dict, dr->dict_size);
                if (rc == ZXC_OK && dr->dict_huf) rc = zxc_mi355x_memcpy_h2d((uint8_t*)b->d_dict + dr->dict_size, dr->dict_huf, ZXC_HUF_TABLE_SIZE);
                s->dict_up = rc == ZXC_OK;
            }
        }
        if (dbg) dbg->up1 = pipe_now_ms();
        if (rc == ZXC_OK)
            rc = zxc_hip_decode_blocks(b->d_comp, (const zxc_dev_job_t*)b->d_jobs, n, b->d_out, (int32_t*)b->d_status, e->block_size, e->verify,
                                       b->d_dict, b->d_dict ? (uint32_t)dr->dict_size : 0u,
                                       (b->d_dict && dr->dict_huf) ? (uint8_t*)b->d_dict + dr->dict_size : NULL, 0u, stream);
    }
    if (rc == ZXC_OK) rc = zxc_hip_memcpy_d2h_async(a->h_st, b->d_status, (size_t)n * sizeof(int32_t), stream);
    if (rc == ZXC_OK) rc = zxc_hip_event_record(a->ev, stream);
    if (dbg) dbg->launched = pipe_now_ms();
    return rc;
}

/* A producer: takes the next piece from the source when its slot is free (one producer at a time, in order), then uploads and
 * enqueues it on its own stream while the other producer is already asking the source for the piece behind it. */
static void* pipe_producer_main(void* arg) {
    pipe_t* e = (pipe_t*)arg;
    if (zxc_mi355x_set_device(e->device) != ZXC_OK) {
        pthread_mutex_lock(&e->mu);
        e->src_done = 1; /* (the consumer then finds no piece; pipe_run reports the failure through the first slot) */
        e->slot[0].rc = ZXC_ERROR_GPU_UNAVAILABLE;
        pthread_cond_broadcast(&e->cv);
        pthread_mutex_unlock(&e->mu);
        return NULL;
    }
    for (;;) {
        pthread_mutex_lock(&e->src_mu);
        pthread_mutex_lock(&e->mu);
        while (!e->stop && !e->src_done && e->next - e->consumed >= (uint64_t)e->k) pthread_cond_wait(&e->cv, &e->mu);
        const int quit = e->stop || e->src_done;
        const uint64_t i = e->next;
        pthread_mutex_unlock(&e->mu);
        if (quit) { pthread_mutex_unlock(&e->src_mu); break; }
        pipe_slot_t* s = &e->slot[i % (uint64_t)e->k];
        if (g_pipe_dbg && i < PIPE_DBG_MAX) g_pipe_dbg[i].src0 = pipe_now_ms();
        const int r = e->source(e->source_ctx, pipe_piece_blocks(e, i), &s->p);
        pthread_mutex_lock(&e->mu);
        if (r == 1) e->next = i + 1;
        else {
            e->src_done = 1;
            if (r < 0) { s->rc = r; s->ready = i + 1; e->next = i + 1; } /* an error the consumer meets in its turn: behind every piece in front of it */
            pthread_cond_broadcast(&e->cv);
        }
        pthread_mutex_unlock(&e->mu);
        pthread_mutex_unlock(&e->src_mu);
        if (r != 1) break;
        const int rc = pipe_enqueue(e, i);
        pthread_mutex_lock(&e->mu);
        s->rc = rc;
        s->ready = i + 1;
        if (rc != ZXC_OK) e->src_done = 1;
        pthread_cond_broadcast(&e->cv);
        pthread_mutex_unlock(&e->mu);
        if (rc != ZXC_OK) break;
    }
    return NULL;
}

/* Runs the pipeline until the source is exhausted (0), a sink stops it (its non-zero verdict) or something fails (< 0).
 * total_comp / total_out: what the whole call can need at most (sizes the slots' reservations); piece_bytes: output per piece
 * after the ramp (0: PIPE_PIECE_BYTES); first_sub: the arena this caller may wait for; want_slots: how many to try to hold. */
static int pipe_run_ex(pipe_source_fn source, void* source_ctx, pipe_sink_fn sink, void* sink_ctx, pipe_enqueue_fn enqueue, void* enqueue_ctx,
                       uint32_t block_size, int verify, const dict_ref_t* dr, size_t total_comp, uint64_t total_out, size_t piece_bytes,
                       size_t first_bytes, int first_sub, int want_slots) {
    if (zxc_mi355x_device_count() <= 0) return ZXC_ERROR_GPU_UNAVAILABLE;
    const int dev = zxc_hip_current_device();
    if (dev < 0 || dev >= HOST_MAX_DEVICES) return ZXC_ERROR_GPU_UNAVAILABLE;
    pthread_once(&g_arena_once, arena_init_all);
    pipe_t* e = (pipe_t*)calloc(1, sizeof *e);
    if (!e) return ZXC_ERROR_MEMORY;
    e->source = source;
    e->source_ctx = source_ctx;
    e->enqueue = enqueue;
    e->enqueue_ctx = enqueue_ctx;
    e->block_size = block_size;
    e->verify = verify;
    e->dr = dr;
    e->device = dev;
    int fixed = 0; /* (the test switch: every piece this size, no ramp) */
    { const char* ev = getenv("ZXC_MI355X_FRAME_BATCH_MIB"); if (ev && atoi(ev) >= 1 && atoi(ev) <= 1024) { piece_bytes = (size_t)atoi(ev) << 20; fixed = 1; } }
    if (piece_bytes == 0) piece_bytes = PIPE_PIECE_BYTES;
    e->max_blocks = (uint32_t)(piece_bytes / block_size);
    if (e->max_blocks < 16u) e->max_blocks = 16u;
    e->first_blocks = fixed ? e->max_blocks : (uint32_t)((first_bytes ? first_bytes : PIPE_FIRST_BYTES) / block_size);
    if (e->first_blocks < 16u) e->first_blocks = 16u;
    if (e->first_blocks > e->max_blocks) e->first_blocks = e->max_blocks;
    { /* a slot's reservation: the largest piece, or the whole call when that is smaller */
        const uint64_t piece_out = (uint64_t)e->max_blocks * block_size, piece_comp = (uint64_t)e->max_blocks * ((uint64_t)block_size + 16u);
        e->cap_out = (size_t)(total_out + block_size < piece_out ? total_out + block_size : piece_out);
        e->cap_comp = (size_t)((uint64_t)total_comp < piece_comp ? (uint64_t)total_comp : piece_comp);
        const uint64_t all_blocks = total_out / block_size + 1u;
        e->cap_blocks = all_blocks < e->max_blocks ? (uint32_t)all_blocks : e->max_blocks;
    }
    const int one_piece = total_out <= (uint64_t)e->first_blocks * block_size;
    if (one_piece || want_slots < 1) want_slots = 1;
    if (want_slots > PIPE_SLOTS) want_slots = PIPE_SLOTS;
    /* arenas: any free one first, else wait for `first_sub`; the others only if nobody holds them */
    int held[ARENA_SUBS];
    memset(held, 0, sizeof held);
    for (int t = 0; t < ARENA_SUBS && e->k == 0; t++) {
        const int sub = (first_sub + t) % ARENA_SUBS;
        if (pthread_mutex_trylock(&g_arena[dev][sub].mu) == 0) { e->slot[e->k++].a = &g_arena[dev][sub]; held[sub] = 1; }
    }
    if (e->k == 0) { pthread_mutex_lock(&g_arena[dev][first_sub % ARENA_SUBS].mu); e->slot[e->k++].a = &g_arena[dev][first_sub % ARENA_SUBS]; held[first_sub % ARENA_SUBS] = 1; }
    for (int sub = 0; sub < ARENA_SUBS && e->k < want_slots; sub++)
        if (!held[sub] && pthread_mutex_trylock(&g_arena[dev][sub].mu) == 0) { e->slot[e->k++].a = &g_arena[dev][sub]; held[sub] = 1; }
    int ret = 0;
    for (int k = 0; k < e->k; k++) {
        e->slot[k].p.jobs = (zxc_dev_job_t*)malloc((size_t)e->max_blocks * sizeof(zxc_dev_job_t));
        if (!e->slot[k].p.jobs) ret = ZXC_ERROR_MEMORY;
    }
    const double t_begin = pipe_now_ms();
    if (getenv("ZXC_MI355X_DEBUG_TIMES") && !g_pipe_dbg) g_pipe_dbg = (pipe_dbg_t*)calloc(PIPE_DBG_MAX, sizeof(pipe_dbg_t));
    if (g_pipe_dbg) memset(g_pipe_dbg, 0, PIPE_DBG_MAX * sizeof(pipe_dbg_t));
    /* producers: two with three or four slots (one piece in the sink, one or two being made), one with two slots */
    pthread_t th[PIPE_PRODUCERS];
    int started = 0;
    for (int k = 0; k < e->k && ret == 0; k++) /* (a stream that cannot be created: that slot's pieces run on the default stream) */
        if (!e->slot[k].a->stream && zxc_hip_stream_create(&e->slot[k].a->stream) != ZXC_OK) e->slot[k].a->stream = NULL;
    if (ret == 0 && e->k >= 2) {
        e->producers = e->k >= 3 ? PIPE_PRODUCERS : 1;
        pthread_mutex_init(&e->mu, NULL);
        pthread_mutex_init(&e->src_mu, NULL);
        pthread_cond_init(&e->cv, NULL);
        for (int q = 0; q < e->producers; q++) {
            if (pthread_create(&th[started], NULL, pipe_producer_main, e) == 0) started++;
            else break;
        }
        if (started == 0) { pthread_mutex_destroy(&e->mu); pthread_mutex_destroy(&e->src_mu); pthread_cond_destroy(&e->cv); }
    }
    if (ret == 0 && started) {
        for (uint64_t i = 0; ret == 0; i++) {
            pipe_slot_t* s = &e->slot[i % (uint64_t)e->k];
            pthread_mutex_lock(&e->mu);
            while (s->ready != i + 1 && !(e->src_done && i >= e->next)) pthread_cond_wait(&e->cv, &e->mu);
            const int have = s->ready == i + 1;
            pthread_mutex_unlock(&e->mu);
            if (!have) { if (i == 0 && e->slot[0].rc < 0) ret = e->slot[0].rc; break; }
            if (s->rc < 0) { ret = s->rc; break; }
            ret = zxc_hip_event_synchronize(s->a->ev);
            if (g_pipe_dbg && i < PIPE_DBG_MAX) g_pipe_dbg[i].ready = pipe_now_ms();
            if (ret == 0) ret = sink(sink_ctx, &s->p, s->a->h_st, &s->b);
            if (g_pipe_dbg && i < PIPE_DBG_MAX) g_pipe_dbg[i].sunk = pipe_now_ms();
            pthread_mutex_lock(&e->mu);
            e->consumed = i + 1;
            pthread_cond_broadcast(&e->cv);
            pthread_mutex_unlock(&e->mu);
        }
        pthread_mutex_lock(&e->mu);
        e->stop = 1;
        pthread_cond_broadcast(&e->cv);
        pthread_mutex_unlock(&e->mu);
        for (int q = 0; q < started; q++) pthread_join(th[q], NULL);
        pthread_mutex_destroy(&e->mu);
        pthread_mutex_destroy(&e->src_mu);
        pthread_cond_destroy(&e->cv);
    } else if (ret == 0) { /* one slot, in series on this thread */
        e->k = 1; /* (the arenas beyond the first are released below all the same) */
        for (uint64_t i = 0; ret == 0; i++) {
            if (g_pipe_dbg && i < PIPE_DBG_MAX) g_pipe_dbg[i].src0 = pipe_now_ms();
            const int r = e->source(e->source_ctx, pipe_piece_blocks(e, i), &e->slot[0].p);
            if (r <= 0) { ret = r; break; }
            ret = pipe_enqueue(e, i);
            if (ret == 0) ret = zxc_hip_event_synchronize(e->slot[0].a->ev);
            if (g_pipe_dbg && i < PIPE_DBG_MAX) g_pipe_dbg[i].ready = pipe_now_ms();
            if (ret == 0) ret = sink(sink_ctx, &e->slot[0].p, e->slot[0].a->h_st, &e->slot[0].b);
            if (g_pipe_dbg && i < PIPE_DBG_MAX) g_pipe_dbg[i].sunk = pipe_now_ms();
        }
    }
    /* nothing of this call may still run on the arenas when they go back */
    for (int sub = 0; sub < ARENA_SUBS; sub++)
        if (held[sub]) (void)zxc_mi355x_synchronize(g_arena[dev][sub].stream);
    if (g_pipe_dbg) {
        fprintf(stderr, "[pipe] %d slot(s), %d producer thread(s), pieces of %u..%u blocks; ms since the call began:\n", e->k, started, e->first_blocks, e->max_blocks);
        for (int i = 0; i < PIPE_DBG_MAX && g_pipe_dbg[i].n; i++)
            fprintf(stderr, "[pipe]  piece %2d: %5u blocks %6.1f MiB in %6.1f MiB out | source %.2f, upload %.2f-%.2f, launched %.2f | ready %.2f, sunk %.2f\n", i,
                    g_pipe_dbg[i].n, g_pipe_dbg[i].comp / 1048576.0, g_pipe_dbg[i].out / 1048576.0, g_pipe_dbg[i].src0 - t_begin, g_pipe_dbg[i].up0 - t_begin,
                    g_pipe_dbg[i].up1 - t_begin, g_pipe_dbg[i].launched - t_begin, g_pipe_dbg[i].ready - t_begin, g_pipe_dbg[i].sunk - t_begin);
        fprintf(stderr, "[pipe]  done at %.2f ms\n", pipe_now_ms() - t_begin);
    }
    for (int sub = 0; sub < ARENA_SUBS; sub++)
        if (held[sub]) {
            arena_t* a = &g_arena[dev][sub];
            for (int w = 0; w < AR_N; w++)
                if (a->buf[w].cap > ARENA_KEEP_MAX) { zxc_mi355x_free(a->buf[w].p); a->buf[w].p = NULL; a->buf[w].cap = 0; }
            pthread_mutex_unlock(&a->mu);
        }
    for (int k = 0; k < PIPE_SLOTS; k++) { free(e->slot[k].p.jobs); free(e->slot[k].p.stage); }
    free(e);
    return ret;
}

static int pipe_run(pipe_source_fn source, void* source_ctx, pipe_sink_fn sink, void* sink_ctx, uint32_t block_size, int verify,
                    const dict_ref_t* dr, size_t total_comp, uint64_t total_out, size_t piece_bytes, int first_sub, int want_slots) {
    return pipe_run_ex(source, source_ctx, sink, sink_ctx, NULL, NULL, block_size, verify, dr, total_comp, total_out, piece_bytes, 0, first_sub, want_slots);
}

/* ------------------------------------------------------------ zxc_decompress */
/* The 8-byte block headers are walked on the host into job tables, one piece at a time: device memory is O(piece), not a
 * function of the untrusted block count, and decoding stops at the first failing or overflowing block like the reference's
 * sequential loop (zxc_dispatch.c:912-1001). A problem found at block k is only reported if blocks 0..k-1 all decode (first
 * failure in stream order wins). */
typedef struct {
    const uint8_t* src;
    size_t src_size, ip;
    uint32_t block_size;
    int file_ck, verify;
    int tail_err;         /* error to report after all queued blocks succeed */
    int saw_eof, done;
    uint32_t global_hash; /* rotl1-xor fold of the stored per-block checksums (zxc_internal.h:1390-1393) */
    size_t stop_ip;       /* != 0: a piece ends in front of this position (the block-by-block walk over an irregular piece stops where the piece did) */
} frame_walk_t;
static int frame_source(void* ctx, uint32_t max_blocks, pipe_piece_t* p) {
    frame_walk_t* w = (frame_walk_t*)ctx;
    zxc_dev_job_t* jobs = p->jobs;
    uint32_t n = 0;
    const size_t span0 = w->ip;
    const uint32_t hash0 = w->global_hash;
    while (!w->done && n < max_blocks) {
        if (w->ip >= w->src_size) { w->done = 1; break; }
        if (w->stop_ip && w->ip >= w->stop_ip) break;
        const size_t rem = w->src_size - w->ip;
        uint8_t type;
        uint32_t csz;
        if (read_block_header(w->src + w->ip, rem, &type, &csz) != ZXC_OK) { w->tail_err = ZXC_ERROR_BAD_HEADER; w->done = 1; break; }
        if (type == BLK_EOF) {
            if (csz != 0) w->tail_err = ZXC_ERROR_BAD_HEADER;
            w->saw_eof = 1;
            w->done = 1;
            break;
        }
        const uint64_t phys = (uint64_t)BLK_HDR + csz + (w->file_ck ? 4u : 0u);
        jobs[n].comp_off = w->ip - span0;
        /* the wrapper sees "all remaining bytes"; any size >= the physical block is equivalent */
        { const uint64_t cs = phys < rem ? phys : rem; jobs[n].comp_size = cs > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cs; }
        jobs[n].out_off = (uint64_t)n * w->block_size;
        jobs[n].out_len = w->block_size;
        n++;
        if (w->verify && phys <= rem)
            w->global_hash = ((w->global_hash << 1) | (w->global_hash >> 31)) ^ rd32(w->src + w->ip + BLK_HDR + csz);
        if (phys >= rem) { w->ip = w->src_size; w->done = 1; break; }
        w->ip += (size_t)phys;
    }
    if (n == 0) return 0;
    p->h_comp = w->src + span0;
    p->comp_bytes = w->ip - span0;
    p->n = n;
    p->out_bytes = (size_t)n * w->block_size;
    /* the walk in front of and behind this piece: the caller restarts from either when a piece turns out irregular */
    p->cookie[0] = span0;
    p->cookie[1] = hash0;
    p->cookie[2] = w->ip;
    p->cookie[3] = w->global_hash;
    p->cookie[4] = (uint64_t)w->done | ((uint64_t)w->saw_eof << 1);
    p->cookie[5] = (uint64_t)(uint32_t)w->tail_err;
    return 1;
}
typedef struct {
    uint8_t* dst;
    size_t dst_capacity, total;
    uint32_t block_size;
    uint64_t irr[6]; /* cookie of the piece that turned out irregular */
} frame_sink_t;
static int frame_sink(void* ctx, const pipe_piece_t* p, const int32_t* st, const dev_bufs_t* b) {
    frame_sink_t* k = (frame_sink_t*)ctx;
    /* sequential semantics: first failing block wins; sizes accumulate in order. A block that is not the frame's last and
     * decodes to another size than block_size makes the frame irregular (legal, never produced by the reference encoder):
     * blocks no longer sit back to back in the slot layout. */
    const int last_piece = (int)(p->cookie[4] & 1u);
    int regular = 1;
    size_t piece_total = 0;
    for (uint32_t i = 0; i < p->n; i++) {
        if (st[i] < 0) return st[i];
        if ((size_t)st[i] > k->dst_capacity - k->total - piece_total) return ZXC_ERROR_DST_TOO_SMALL;
        if ((uint32_t)st[i] != k->block_size && !(last_piece && i + 1 == p->n)) regular = 0;
        if ((uint32_t)st[i] > k->block_size) regular = 0;
        piece_total += (size_t)st[i];
    }
    if (!regular) { memcpy(k->irr, p->cookie, sizeof k->irr); return PIPE_IRREGULAR; }
    const int rc = zxc_mi355x_memcpy_d2h(k->dst + k->total, b->d_out, piece_total);
    if (rc != ZXC_OK) return rc;
    k->total += piece_total;
    return 0;
}
/* An irregular piece: decoded sizes are a property of the blocks alone — run it again with one cap-sized slot per block and
 * gather block by block (in series: this path exists for correctness, the reference encoder never writes such frames). */
static int frame_irregular_piece(frame_walk_t* w, frame_sink_t* k, const dict_ref_t* dr, uint32_t max_blocks) {
    const uint32_t slot = (w->block_size + TAIL_PAD + 15u) & ~15u;
    pipe_piece_t p;
    memset(&p, 0, sizeof p);
    p.jobs = (zxc_dev_job_t*)malloc((size_t)max_blocks * sizeof(zxc_dev_job_t));
    int32_t* st = (int32_t*)malloc((size_t)max_blocks * sizeof(int32_t));
    int rc = (p.jobs && st) ? ZXC_OK : ZXC_ERROR_MEMORY;
    if (rc == ZXC_OK && frame_source(w, max_blocks, &p) != 1) rc = ZXC_ERROR_CORRUPT_DATA; /* (cannot happen: the walk was rewound to a piece) */
    dev_bufs_t b;
    memset(&b, 0, sizeof b);
    if (rc == ZXC_OK) {
        for (uint32_t i = 0; i < p.n; i++) { p.jobs[i].out_off = (uint64_t)i * slot; p.jobs[i].out_len = slot; }
        rc = run_jobs_on(p.h_comp, p.comp_bytes, p.jobs, p.n, (size_t)p.n * slot, w->block_size, 0u, w->verify, st, &b, dr, 0, NULL);
        if (rc != ZXC_OK) memset(&b, 0, sizeof b);
    }
    for (uint32_t i = 0; i < p.n && rc == ZXC_OK; i++) {
        /* a status is never trusted as a copy length */
        if (st[i] < 0) { rc = st[i]; break; }
        if ((size_t)st[i] > k->dst_capacity - k->total) { rc = ZXC_ERROR_DST_TOO_SMALL; break; }
        if ((uint32_t)st[i] > slot) { rc = ZXC_ERROR_CORRUPT_DATA; break; }
        rc = zxc_mi355x_memcpy_d2h(k->dst + k->total, (const uint8_t*)b.d_out + (size_t)i * slot, (size_t)st[i]);
        if (rc == ZXC_OK) k->total += (size_t)st[i];
    }
    dev_bufs_free(&b);
    free(p.jobs);
    free(st);
    return rc;
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

    frame_walk_t w;
    memset(&w, 0, sizeof w);
    w.src = src;
    w.src_size = src_size;
    w.ip = ZXC_FILE_HEADER_SIZE;
    w.block_size = block_size;
    w.file_ck = file_ck;
    w.verify = verify;
    frame_sink_t k;
    memset(&k, 0, sizeof k);
    k.dst = dst;
    k.dst_capacity = dst_capacity;
    k.block_size = block_size;
    /* (the footer's size is a hint for the slots' reservations only: bounded by the caller's capacity) */
    uint64_t out_hint = rd64(src + src_size - ZXC_FILE_FOOTER_SIZE);
    if (out_hint > (uint64_t)dst_capacity) out_hint = dst_capacity;
    for (;;) {
        const int r = pipe_run(frame_source, &w, frame_sink, &k, block_size, verify, &dr, src_size, out_hint, 0, 0, PIPE_DECODE_SLOTS);
        if (r < 0) return r;
        if (r != PIPE_IRREGULAR) break;
        /* rewind the walk (it ran ahead of the sink) to the front of the irregular piece, take that piece block by block ... */
        w.ip = (size_t)k.irr[0];
        w.global_hash = (uint32_t)k.irr[1];
        w.done = 0;
        w.saw_eof = 0;
        w.tail_err = 0;
        uint32_t nb = (uint32_t)(PIPE_PIECE_BYTES / block_size); /* (at most what a piece can hold: the walk stops where the piece did) */
        { const char* ev = getenv("ZXC_MI355X_FRAME_BATCH_MIB"); if (ev && atoi(ev) >= 1 && atoi(ev) <= 1024) nb = (uint32_t)(((size_t)atoi(ev) << 20) / block_size); }
        if (nb < 16u) nb = 16u;
        /* the piece had exactly this many blocks or fewer: walk block by block up to where it ended */
        w.stop_ip = (size_t)k.irr[2];
        while (w.ip < (size_t)k.irr[2] && !w.done) {
            const int rc = frame_irregular_piece(&w, &k, &dr, nb);
            if (rc != ZXC_OK) return rc;
        }
        w.stop_ip = 0;
        /* ... and go on behind it */
        if (w.done) break;
    }
    if (w.tail_err) return w.tail_err;
    if (w.saw_eof) { /* footer: stored size must equal what was produced (zxc_dispatch.c:936-943) */
        const uint8_t* footer = src + src_size - ZXC_FILE_FOOTER_SIZE;
        if (rd64(footer) != (uint64_t)k.total) return ZXC_ERROR_CORRUPT_DATA;
        if (verify && rd32(footer + 8) != w.global_hash) return ZXC_ERROR_BAD_CHECKSUM; /* :945-952 */
    }
    return (int64_t)k.total;
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

/* zxc_compress over pieces of blocks, through the piece pipeline above (round 5): two producer threads upload the source pieces
 * and launch their encodes (a piece's encode takes ~3.5 ms whatever its size — one round of wavefronts — so three pieces sit in
 * flight and fill the chip), the calling thread turns a finished piece's block sizes into offsets, compacts its slots
 * (zxc_gather_blocks_kernel) and brings the bytes back; device buffers live in the staging arenas across calls. Round 4: one helper
 * thread, upload and launch of batch i + 1 in series beside the download of batch i, 3 ms of hipMalloc / hipFree per call:
 * 1 GiB in 41-44 ms against 28 ms of encode launches. Archives are byte for byte the one-shot ones whatever the piece size
 * (tests/test_gpu_encode.py); ZXC_MI355X_DEBUG_TIMES=1 prints the pieces' timeline. */
typedef struct {
    const uint8_t* src;
    size_t src_size, block_size;
    uint32_t next, nb;
} comp_src_t;
static int comp_source(void* ctx, uint32_t max_blocks, pipe_piece_t* p) {
    comp_src_t* c = (comp_src_t*)ctx;
    if (c->next >= c->nb) return 0;
    const uint32_t f = c->next, n = c->nb - f < max_blocks ? c->nb - f : max_blocks;
    const size_t o = (size_t)f * c->block_size;
    p->h_comp = c->src + o; /* (here: the SOURCE bytes of the piece's blocks) */
    p->comp_bytes = c->src_size - o < (size_t)n * c->block_size ? c->src_size - o : (size_t)n * c->block_size;
    p->n = n;
    p->out_bytes = (size_t)n * (c->block_size + 64);
    p->cookie[0] = f;
    c->next = f + n;
    return 1;
}
typedef struct { int level, checksum; } comp_enq_t;
/* a slot's buffers in dev_bufs_t terms: d_comp = the source piece, d_dict = the encoder's per-block slots, d_status = block sizes,
 * d_jobs = the blocks' offsets in the compacted output, d_out = that output */
static int comp_enqueue(struct pipe_s* e, pipe_slot_t* s, void* stream) {
    const comp_enq_t* c = (const comp_enq_t*)e->enqueue_ctx;
    arena_t* a = s->a;
    dev_bufs_t* b = &s->b;
    const uint32_t stride = zxc_mi355x_encode_slot_stride(e->block_size);
    b->d_comp = arena_reserve(a, AR_COMP, (size_t)e->cap_blocks * e->block_size + 64);
    b->d_dict = arena_reserve(a, AR_SLOTS, (size_t)e->cap_blocks * stride);
    b->d_status = arena_reserve(a, AR_STATUS, (size_t)e->max_blocks * sizeof(int32_t));
    b->d_jobs = arena_reserve(a, AR_JOBS, (size_t)e->max_blocks * sizeof(zxc_dev_job_t));
    b->d_out = arena_reserve(a, AR_OUT, (size_t)e->cap_blocks * ((size_t)e->block_size + 72) + 64); /* (a block is never larger than RAW: header + bytes + trailer) */
    if (!b->d_comp || !b->d_dict || !b->d_status || !b->d_jobs || !b->d_out) return ZXC_ERROR_MEMORY;
    int rc = zxc_hip_memcpy_h2d_async(b->d_comp, s->p.h_comp, s->p.comp_bytes, stream);
    if (rc == ZXC_OK)
        rc = zxc_mi355x_encode_blocks_device(b->d_comp, s->p.comp_bytes, e->block_size, c->level, c->checksum, b->d_dict, (uint32_t*)b->d_status, stream);
    /* the compaction follows on the same stream, offsets computed on the device: behind three pieces' encodes a launch from the
     * sink (round 4: sizes to the host, offsets back, gather on the default stream) waited milliseconds for a free compute unit */
    if (rc == ZXC_OK) rc = zxc_hip_block_offsets((uint32_t*)b->d_status, (uint64_t*)b->d_jobs, s->p.n, e->block_size + 65u, stream);
    if (rc == ZXC_OK)
        rc = zxc_mi355x_gather_blocks_device(b->d_dict, e->block_size, (const uint32_t*)b->d_status, (const uint64_t*)b->d_jobs, b->d_out, s->p.n, stream);
    return rc;
}
typedef struct {
    uint8_t* dst;
    size_t dst_capacity, op, tail_need, block_size;
    uint32_t* sizes; /* [nb]: the seek table's entries */
    uint64_t* offs;  /* scratch, one piece */
    uint32_t global_hash;
    int checksum;
} comp_sink_t;
static int comp_sink(void* ctx, const pipe_piece_t* p, const int32_t* st, const dev_bufs_t* b) {
    comp_sink_t* k = (comp_sink_t*)ctx;
    const uint32_t f = (uint32_t)p->cookie[0], n = p->n;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t sz = (uint32_t)st[i];
        /* (never trusted as a copy length, nor as the place of the trailer the global hash is folded from: a block is at least its
         *  8-byte header, + 4 with checksums) */
        if (sz > k->block_size + 64 || sz < 8u + (k->checksum ? 4u : 0u)) return ZXC_ERROR_CORRUPT_DATA;
        k->sizes[f + i] = sz;
        k->offs[i] = total;
        total += sz;
    }
    if ((uint64_t)k->op + total + k->tail_need > (uint64_t)k->dst_capacity) return ZXC_ERROR_DST_TOO_SMALL;
    /* (the piece's blocks sit compacted in d_out: comp_enqueue) */
    const int rc = zxc_mi355x_memcpy_d2h(k->dst + k->op, b->d_out, (size_t)total);
    if (rc != ZXC_OK) return rc;
    if (k->checksum) /* fold the block trailers in stream order (zxc_dispatch.c:754-759) */
        for (uint32_t i = 0; i < n; i++)
            k->global_hash = ((k->global_hash << 1) | (k->global_hash >> 31)) ^ rd32(k->dst + k->op + k->offs[i] + k->sizes[f + i] - 4);
    k->op += (size_t)total;
    return 0;
}
#define COMP_PIECE_BYTES ((size_t)64 << 20)
/* -> ZXC_OK and *op_io advanced over the nb encoded blocks, sizes[] and *hash_io filled; or a negative zxc_error_t */
static int compress_pieces(const uint8_t* src, size_t src_size, size_t block_size, int level, int checksum_enabled, size_t tail_need,
                           uint8_t* dst, size_t dst_capacity, size_t* op_io, uint32_t* sizes, uint32_t nb, size_t piece_bytes, uint32_t* hash_io) {
    comp_src_t cs = {src, src_size, block_size, 0, nb};
    comp_enq_t ce = {level, checksum_enabled};
    comp_sink_t ck;
    memset(&ck, 0, sizeof ck);
    ck.dst = dst;
    ck.dst_capacity = dst_capacity;
    ck.op = *op_io;
    ck.tail_need = tail_need;
    ck.block_size = block_size;
    ck.sizes = sizes;
    ck.global_hash = *hash_io;
    ck.checksum = checksum_enabled;
    uint32_t piece_blocks = (uint32_t)(piece_bytes / block_size);
    if (piece_blocks < 16u) piece_blocks = 16u;
    ck.offs = (uint64_t*)malloc((size_t)piece_blocks * sizeof(uint64_t));
    if (!ck.offs) return ZXC_ERROR_MEMORY;
    /* (a short ramp — 16, 32, 64 MiB — so that the first encodes start while the link is still busy with the uploads behind them;
     *  six slots: a piece's encode takes ~3.5 ms whatever its size, five of them in flight keep the chip full across their tails) */
    const int rc = pipe_run_ex(comp_source, &cs, comp_sink, &ck, comp_enqueue, &ce, (uint32_t)block_size, 0, NULL, src_size, (uint64_t)nb * block_size,
                               (size_t)piece_blocks * block_size, (size_t)16 << 20, 0, PIPE_SLOTS);
    free(ck.offs);
    if (rc != 0) return rc < 0 ? rc : ZXC_ERROR_CORRUPT_DATA;
    *op_io = ck.op;
    *hash_io = ck.global_hash;
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
    size_t batch_bytes = COMP_PIECE_BYTES;
    { const char* e = getenv("ZXC_MI355X_FRAME_BATCH_MIB"); if (e && atoi(e) >= 1 && atoi(e) <= 1024) batch_bytes = (size_t)atoi(e) << 20; }
    const uint32_t batch_blocks = (uint32_t)(batch_bytes / block_size > 16 ? batch_bytes / block_size : 16);
    if (nb > 0 && !dict_size) { /* through the piece pipeline: pipelined when there is more than one piece, else in series on this thread —
                                 * in the staging arenas either way (round 4: five hipMalloc + hipFree = ~3 ms per call up to 64 MiB) */
        if (zxc_mi355x_device_count() <= 0) return ZXC_ERROR_GPU_UNAVAILABLE;
        sizes = (uint32_t*)malloc((size_t)nb * sizeof(uint32_t));
        if (!sizes) return ZXC_ERROR_MEMORY;
        const size_t tail_need = BLK_HDR + (seekable ? zxc_seek_table_size(nb) : 0) + ZXC_FILE_FOOTER_SIZE;
        const int rc = compress_pieces((const uint8_t*)src, src_size, block_size, level, checksum_enabled, tail_need, dst, dst_capacity,
                                       &op, sizes, nb, (size_t)batch_blocks * block_size, &global_hash);
        if (rc != ZXC_OK) { free(sizes); return rc; }
    } else if (nb > 0) { /* with a dictionary: one shot */
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

/* Bytes [offset, offset + len) of the archive into dst, on the calling thread's device: the covered blocks go through the piece
 * pipeline above (device memory is O(piece) whatever the range; upload of piece i+1 | decode | download of piece i), each piece
 * = one launch over its blocks' compressed span. `sub`: the staging arena this caller may wait for; `slots`: how many it tries
 * to hold. */
typedef struct {
    zxc_seekable* s;
    uint32_t next, b1;
} range_src_t;
static int range_source(void* ctx, uint32_t max_blocks, pipe_piece_t* p) {
    range_src_t* r = (range_src_t*)ctx;
    zxc_seekable* s = r->s;
    if (r->next > r->b1) return 0;
    const uint32_t f = r->next;
    const uint32_t n = (r->b1 - f + 1u) < max_blocks ? (r->b1 - f + 1u) : max_blocks;
    const uint64_t c0 = s->comp_offsets[f], c1 = s->comp_offsets[f + n];
    if (c1 > s->src_size) return ZXC_ERROR_SRC_TOO_SMALL;
    const size_t comp_bytes = (size_t)(c1 - c0);
    /* compressed span of the piece's blocks: borrowed buffer or reader callback (into the slot's host staging) */
    if (s->src) {
        p->h_comp = s->src + c0;
    } else {
        if (p->stage_cap < comp_bytes) {
            free(p->stage);
            p->stage = (uint8_t*)malloc(comp_bytes ? comp_bytes : 1);
            p->stage_cap = p->stage ? comp_bytes : 0;
            if (!p->stage) return ZXC_ERROR_MEMORY;
        }
        const int64_t got = s->reader.read_at(s->reader.ctx, p->stage, comp_bytes, c0);
        if (got != (int64_t)comp_bytes) return got < 0 ? (int)got : ZXC_ERROR_IO;
        p->h_comp = p->stage;
    }
    p->comp_bytes = comp_bytes;
    zxc_mi355x_plan_seekable(s, f, n, c0, p->jobs);
    /* The reference decodes each block with cap block_size + 2112 and keeps what the range needs (zxc_seekable.c:758-780):
     * keep whole slots here. */
    for (uint32_t k = 0; k < n; k++) p->jobs[k].out_len = s->block_size;
    p->n = n;
    p->out_bytes = (size_t)n * s->block_size;
    p->cookie[0] = f;
    r->next = f + n;
    return 1;
}
typedef struct {
    zxc_seekable* s;
    uint8_t* dst;
    uint64_t offset;
    size_t len;
} range_sink_t;
static int range_sink(void* ctx, const pipe_piece_t* p, const int32_t* st, const dev_bufs_t* b) {
    range_sink_t* k = (range_sink_t*)ctx;
    zxc_seekable* s = k->s;
    const uint32_t f = (uint32_t)p->cookie[0], n = p->n;
    /* first failing block in job order wins (zxc_seekable.c:1097-1104) */
    const uint64_t lo = (uint64_t)f * s->block_size; /* decoded position of the piece's first byte */
    const uint64_t from = k->offset > lo ? k->offset : lo;
    const uint64_t piece_end = lo + (uint64_t)n * s->block_size;
    const uint64_t to = (k->offset + k->len) < piece_end ? (k->offset + k->len) : piece_end;
    for (uint32_t i = 0; i < n; i++) {
        if (st[i] < 0) return st[i];
        /* a block that decodes short of what the range needs from it */
        const uint64_t bstart = lo + (uint64_t)i * s->block_size;
        const uint64_t bend = bstart + zxc_seekable_get_block_decomp_size(s, f + i);
        const uint64_t bneed_hi = to < bend ? to : bend;
        if (bneed_hi > bstart && (uint64_t)st[i] < bneed_hi - bstart) return ZXC_ERROR_CORRUPT_DATA;
    }
    if (to > from) return zxc_mi355x_memcpy_d2h(k->dst + (size_t)(from - k->offset), (const uint8_t*)b->d_out + (size_t)(from - lo), (size_t)(to - from));
    return 0;
}
static int64_t seek_range_on(zxc_seekable* s, uint8_t* dst, const uint64_t offset, const size_t len, int sub, int slots) {
    const dict_ref_t dr = {s->dict, s->dict_size, s->has_dict_huf ? s->dict_huf : NULL};
    const uint32_t b0 = (uint32_t)(offset / s->block_size);
    const uint32_t b1 = (uint32_t)((offset + len - 1) / s->block_size);
    range_src_t src = {s, b0, b1};
    range_sink_t snk = {s, dst, offset, len};
    const uint64_t comp_span = s->comp_offsets[b1 + 1u] - s->comp_offsets[b0];
    const int rc = pipe_run(range_source, &src, range_sink, &snk, s->block_size, 0, &dr, (size_t)comp_span,
                            (uint64_t)(b1 - b0 + 1u) * s->block_size, 0, sub, slots);
    return rc < 0 ? (int64_t)rc : (int64_t)len;
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
    return seek_range_on(s, (uint8_t*)dst, offset, len, 0, PIPE_DECODE_SLOTS);
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
    if (zxc_mi355x_set_device(a->device) != ZXC_OK) {
        a->result = ZXC_ERROR_GPU_UNAVAILABLE;
        return NULL;
    }
    /* (each part runs its own piece pipeline: its own stream, the arena of its `sub` + one more if one is free) */
    a->result = seek_range_on(a->s, a->dst, a->offset, a->len, a->sub, 2);
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
/* `in_workspace`: the handle lives in a caller-supplied workspace (Static Context API below): zxc_free_* leave it alone, the
 * block size is locked; `dense`: carved for levels 6-7 */
struct zxc_cctx_s { int level; size_t block_size; int checksum; int in_workspace; int dense; };
struct zxc_dctx_s { int in_workspace; size_t block_size; };

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
void zxc_free_cctx(zxc_cctx* cctx) { if (cctx && !cctx->in_workspace) free(cctx); }
zxc_dctx* zxc_create_dctx(void) { return (zxc_dctx*)calloc(1, sizeof(zxc_dctx)); }
void zxc_free_dctx(zxc_dctx* dctx) { if (dctx && !dctx->in_workspace) free(dctx); }

/* Static Context API (reference include/zxc_buffer.h:494-604, src/lib/zxc_dispatch.c:1860-1965): the whole context in ONE buffer
 * the caller owns, so that the library never touches the host allocator for it. The reference carves its hash / chain tables,
 * sequence buffers and (levels 6-7) the optimal parser's scratch out of that buffer; here those live in LDS and in device memory
 * (the staging arenas, sized per call), so what the workspace holds is the handle — one cache line — and, for a context carved
 * at the dense tier, a second line (the reference's sizes grow at level 6 too: callers that size per level keep working). The
 * contract that matters to callers is the reference's: block_size locked (ZXC_ERROR_BAD_BLOCK_SIZE), a raise into levels 6-7 on
 * a workspace carved below them refused (ZXC_ERROR_BAD_LEVEL), level / checksum otherwise per call, zxc_free_* no-ops. */
#define STATIC_LINE 64u
static int valid_block_size(size_t bs) { return bs >= ZXC_BLOCK_SIZE_MIN && bs <= ZXC_BLOCK_SIZE_MAX && !(bs & (bs - 1)); }

size_t zxc_static_cctx_workspace_size(const size_t block_size, const int level) {
    if (!valid_block_size(block_size) || level < ZXC_LEVEL_FASTEST || level > ZXC_LEVEL_ULTRA) return 0;
    return STATIC_LINE * (level >= ZXC_LEVEL_DENSITY ? 2u : 1u);
}

zxc_cctx* zxc_init_static_cctx(void* workspace, const size_t workspace_size, const zxc_compress_opts_t* opts) {
    if (!workspace || !opts || ((uintptr_t)workspace & (_Alignof(zxc_cctx) - 1))) return NULL;
    const int level = opts->level > 0 ? opts->level : ZXC_LEVEL_DEFAULT;
    const size_t bs = opts->block_size > 0 ? opts->block_size : ZXC_BLOCK_SIZE_DEFAULT;
    const size_t need = zxc_static_cctx_workspace_size(bs, level);
    if (need == 0 || workspace_size < need) return NULL;
    zxc_cctx* c = (zxc_cctx*)workspace;
    memset(c, 0, sizeof(*c));
    c->level = level;
    c->block_size = bs;
    c->checksum = opts->checksum_enabled;
    c->in_workspace = 1;
    c->dense = level >= ZXC_LEVEL_DENSITY;
    return c;
}

size_t zxc_static_dctx_workspace_size(const size_t block_size) { return valid_block_size(block_size) ? STATIC_LINE : 0; }

zxc_dctx* zxc_init_static_dctx(void* workspace, const size_t workspace_size, const size_t block_size) {
    if (!workspace || !valid_block_size(block_size) || workspace_size < STATIC_LINE || ((uintptr_t)workspace & (_Alignof(zxc_dctx) - 1)))
        return NULL;
    zxc_dctx* d = (zxc_dctx*)workspace;
    memset(d, 0, sizeof(*d));
    d->in_workspace = 1;
    d->block_size = block_size;
    return d;
}

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
    /* a static context: the block size is locked, the dense tier only if carved for it (src/lib/zxc_dispatch.c:1348-1355) */
    if (cctx->in_workspace && o.block_size != cctx->block_size) return ZXC_ERROR_BAD_BLOCK_SIZE;
    if (cctx->in_workspace && o.level >= ZXC_LEVEL_DENSITY && !cctx->dense) return ZXC_ERROR_BAD_LEVEL;
    cctx->level = o.level;
    cctx->block_size = o.block_size;
    cctx->checksum = o.checksum_enabled;
    return zxc_compress(src, src_size, dst, dst_capacity, &o);
}

int64_t zxc_decompress_dctx(zxc_dctx* dctx, const void* src, const size_t src_size, void* dst, const size_t dst_capacity,
                            const zxc_decompress_opts_t* opts) {
    if (!dctx) return ZXC_ERROR_NULL_INPUT;
    if (dctx->in_workspace) { /* locked to its block size (src/lib/zxc_dispatch.c:1512-1520) */
        if (!src || !dst || src_size < ZXC_FILE_HEADER_SIZE) return ZXC_ERROR_NULL_INPUT;
        uint32_t bs = 0, did = 0;
        int ck = 0;
        if (read_file_header((const uint8_t*)src, src_size, &bs, &ck, &did) != ZXC_OK) return ZXC_ERROR_BAD_HEADER;
        if (bs != dctx->block_size) return ZXC_ERROR_BAD_BLOCK_SIZE;
    }
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
    if (cctx->in_workspace) { /* src/lib/zxc_dispatch.c:1653-1661 */
        const size_t eff = b_dict_size ? block_size_ceil(b_dict_size + bs) : bs;
        if (eff != cctx->block_size) return ZXC_ERROR_BAD_BLOCK_SIZE;
        if (level >= ZXC_LEVEL_DENSITY && !cctx->dense) return ZXC_ERROR_BAD_LEVEL;
    }
    cctx->level = level;
    /* a static context keeps the size it was carved for (the lock above and zxc_compress_cctx's compare against it; the reference stores
     * the EFFECTIVE size there, which for a locked context is that same value: src/lib/zxc_dispatch.c:1662-1667) — storing the base size
     * made the second identical call with a dictionary fail with BAD_BLOCK_SIZE */
    if (!cctx->in_workspace) cctx->block_size = bs;
    cctx->checksum = checksum_enabled;
    if (zxc_mi355x_device_count() <= 0) return ZXC_ERROR_GPU_UNAVAILABLE;
    if (!b_dict_size) { /* one piece of one block in the staging arena (no allocation on the device per call) */
        size_t op = 0;
        uint32_t one_size = 0, unused_hash = 0;
        const int prc = compress_pieces((const uint8_t*)src, src_size, bs, level, checksum_enabled, 0, (uint8_t*)dst, dst_capacity, &op, &one_size, 1u,
                                        COMP_PIECE_BYTES, &unused_hash);
        return prc == ZXC_OK ? (int64_t)op : prc;
    }
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

#include "zxc_stream_host.c"
#include "zxc_pstream_host.c"
#include "zxc_dict_train_host.c"
