/* zxc_pstream_host.c — push streaming (included at the end of zxc_host.c, like zxc_stream_host.inc; a .c so that
 * bench.kernel_sources_hash(), which covers the device sources *.hip / *.inc / *.h of this directory, stays a property of the
 * kernels).
 *
 * Reference: include/zxc_pstream.h:82-292, src/lib/zxc_pstream.c (cstream states :69-79, zxc_cstream_compress :446-508,
 * zxc_cstream_end :525-591; dstream states :634-647, zxc_dstream_decompress :1046-1176). Same entry points, return values and
 * error codes; the difference is the unit of work. The reference calls its block codec once per full block on the calling
 * thread. Here one call hands EVERY block its input completes to one launch:
 *   compress:  [accumulator block (when the previous calls left a partial one and this input fills it)] + all whole blocks of
 *              the caller's input, uploaded side by side, encoded, compacted on the device (offsets kernel + gather) and copied
 *              back — straight into the caller's out when it has the room, else into the pending buffer that later calls drain;
 *   decompress: the block frames that lie whole inside the caller's input are uploaded as ONE span from where they are (no
 *              staging copy), in front of them the one frame that straddled the previous call (the carry buffer); one launch,
 *              statuses checked in stream order (first failing block wins, what precedes it is still delivered), output copied
 *              straight into out when it fits, else staged and drained.
 * A batch never outlives the call that collected it, so "the call returned 0" means what it means in the reference. The device
 * buffers belong to the context (grown to the largest batch seen, at most PS_WINDOW_BYTES of blocks) on the device that was
 * current at its first launch. */
#include "../../include/zxc_pstream.h"

/* source / decoded bytes of one launch: 128 MiB = 2 048 blocks of 64 KiB, one full round of the encoder's workgroups (256 CUs x 8):
 * a launch takes ~3.5 ms whether it holds 512 blocks or 2 048 (measured, profiles/r5p_pstream_bench.log: 8.5 -> 18 GB/s of source).
 * A context's device buffers are about four windows on the compression side, two on the decompression side, grown on demand:
 * a caller that feeds 1 MiB per call never allocates more than that takes. ZXC_MI355X_PSTREAM_WINDOW_MIB = 1 .. 1024 overrides
 * it when a context is created. */
static size_t ps_window_bytes(void) {
    const char* e = getenv("ZXC_MI355X_PSTREAM_WINDOW_MIB");
    if (e && atoi(e) >= 1 && atoi(e) <= 1024) return (size_t)atoi(e) << 20;
    return (size_t)128 << 20;
}
#define PS_WINDOW_BYTES (ps_window_bytes())

static uint32_t ps_window_blocks(size_t block_size) {
    const size_t n = PS_WINDOW_BYTES / block_size;
    return (uint32_t)(n < 16 ? 16 : n);
}
static void* ps_dev_reserve(dbuf_t* d, size_t need) {
    if (d->cap < need) {
        zxc_mi355x_free(d->p);
        d->p = zxc_mi355x_malloc(need);
        d->cap = d->p ? need : 0;
    }
    return d->p;
}
static int ps_host_reserve(uint8_t** p, size_t* cap, size_t need, int keep) {
    if (*cap >= need) return ZXC_OK;
    uint8_t* nb = keep ? (uint8_t*)realloc(*p, need) : (free(*p), (uint8_t*)malloc(need));
    if (!nb) { if (!keep) { *p = NULL; *cap = 0; } return ZXC_ERROR_MEMORY; }
    *p = nb;
    *cap = need;
    return ZXC_OK;
}
/* the context's device: the one current at its first launch; a caller that moved on to another device is switched for the
 * duration of the batch */
static int ps_enter_device(int* ctx_dev, int* prev) {
    if (zxc_mi355x_device_count() <= 0) return ZXC_ERROR_GPU_UNAVAILABLE;
    const int cur = zxc_hip_current_device();
    if (cur < 0) return ZXC_ERROR_GPU_UNAVAILABLE;
    *prev = cur;
    if (*ctx_dev < 0) *ctx_dev = cur;
    return cur == *ctx_dev ? ZXC_OK : zxc_mi355x_set_device(*ctx_dev);
}
static void ps_leave_device(int ctx_dev, int prev) {
    if (prev >= 0 && prev != ctx_dev) (void)zxc_mi355x_set_device(prev);
}

/* ============================================================ compression */
enum { CS_INIT = 0, CS_DRAIN_HEADER, CS_ACCUMULATE, CS_DRAIN_BLOCK, CS_DRAIN_LAST, CS_DRAIN_EOF, CS_DRAIN_FOOTER, CS_DONE, CS_ERRORED };

struct zxc_cstream_s {
    int level, checksum;
    size_t block_size;
    uint32_t max_blocks;  /* blocks per launch */
    uint8_t* in_block;    /* one block: input that does not yet make a whole block */
    size_t in_used;
    uint8_t* pending;     /* output the caller has not drained yet */
    size_t pending_cap, pending_len, pending_pos;
    uint32_t* h_sizes;
    size_t h_sizes_cap;   /* entries */
    uint64_t total_in;
    uint32_t global_hash;
    int state, error_code;
    int dev;
    dbuf_t d_src, d_slots, d_sizes, d_offs, d_out;
};

static int cs_fail(zxc_cstream* cs, int code) {
    cs->error_code = code;
    cs->state = CS_ERRORED;
    return code;
}

zxc_cstream* zxc_cstream_create(const zxc_compress_opts_t* opts) {
    /* dictionaries are refused, not dropped: the push format has no dict_id (src/lib/zxc_pstream.c:275-280) */
    if (opts && (opts->dict || opts->dict_size || opts->dict_huf)) return NULL;
    const size_t bs = (opts && opts->block_size > 0) ? opts->block_size : ZXC_BLOCK_SIZE_DEFAULT;
    if (bs < ZXC_BLOCK_SIZE_MIN || bs > ZXC_BLOCK_SIZE_MAX || (bs & (bs - 1))) return NULL;
    zxc_cstream* cs = (zxc_cstream*)calloc(1, sizeof(*cs));
    if (!cs) return NULL;
    int level = (opts && opts->level > 0) ? opts->level : ZXC_LEVEL_DEFAULT;
    cs->level = level > ZXC_LEVEL_ULTRA ? ZXC_LEVEL_ULTRA : level;
    cs->checksum = opts ? (opts->checksum_enabled != 0) : 0;
    cs->block_size = bs;
    cs->max_blocks = ps_window_blocks(bs);
    cs->dev = -1;
    cs->in_block = (uint8_t*)malloc(bs);
    cs->pending_cap = 64;
    cs->pending = (uint8_t*)malloc(cs->pending_cap);
    if (!cs->in_block || !cs->pending) { zxc_cstream_free(cs); return NULL; }
    cs->state = CS_INIT;
    return cs;
}

void zxc_cstream_free(zxc_cstream* cs) {
    if (!cs) return;
    if (cs->dev >= 0) {
        int prev = -1;
        if (ps_enter_device(&cs->dev, &prev) == ZXC_OK) {
            zxc_mi355x_free(cs->d_src.p); zxc_mi355x_free(cs->d_slots.p); zxc_mi355x_free(cs->d_sizes.p);
            zxc_mi355x_free(cs->d_offs.p); zxc_mi355x_free(cs->d_out.p);
        }
        ps_leave_device(cs->dev, prev);
    }
    free(cs->h_sizes);
    free(cs->pending);
    free(cs->in_block);
    free(cs);
}

size_t zxc_cstream_in_size(const zxc_cstream* cs) { return cs ? (size_t)cs->max_blocks * cs->block_size : 0; }
size_t zxc_cstream_out_size(const zxc_cstream* cs) {
    return cs ? (size_t)cs->max_blocks * (size_t)zxc_compress_block_bound(cs->block_size) : 0;
}

/* One launch over [a[0..alen) | b[0..blen)] (a: the accumulator, one block or the stream's last partial one; b: whole blocks in
 * the caller's input). The blocks' frames land back to back in out (when it has room for all of them) or in pending. */
static int cs_encode(zxc_cstream* cs, const uint8_t* a, size_t alen, const uint8_t* b, size_t blen, zxc_outbuf_t* out) {
    const size_t total = alen + blen, bs = cs->block_size;
    const uint32_t nb = (uint32_t)((total + bs - 1) / bs);
    if (nb == 0) return ZXC_OK;
    int prev = -1;
    int rc = ps_enter_device(&cs->dev, &prev);
    if (rc != ZXC_OK) return rc;
    const uint32_t stride = zxc_mi355x_encode_slot_stride((uint32_t)bs);
    if (cs->h_sizes_cap < nb) {
        free(cs->h_sizes);
        cs->h_sizes = (uint32_t*)malloc((size_t)nb * sizeof(uint32_t));
        cs->h_sizes_cap = cs->h_sizes ? nb : 0;
    }
    uint8_t* d_src = (uint8_t*)ps_dev_reserve(&cs->d_src, total + 64); /* (+64: the match finder's 16-byte compares, zxc_mi355x.h) */
    void* d_slots = ps_dev_reserve(&cs->d_slots, (size_t)nb * stride);
    void* d_sizes = ps_dev_reserve(&cs->d_sizes, (size_t)nb * 4);
    void* d_offs = ps_dev_reserve(&cs->d_offs, (size_t)nb * 8);
    void* d_out = ps_dev_reserve(&cs->d_out, (size_t)nb * (bs + 72) + 64);
    if (!cs->h_sizes || !d_src || !d_slots || !d_sizes || !d_offs || !d_out) rc = ZXC_ERROR_MEMORY;
    if (rc == ZXC_OK && alen) rc = zxc_mi355x_memcpy_h2d(d_src, a, alen);
    if (rc == ZXC_OK && blen) rc = zxc_mi355x_memcpy_h2d(d_src + alen, b, blen);
    if (rc == ZXC_OK) rc = zxc_mi355x_encode_blocks_device(d_src, total, (uint32_t)bs, cs->level, cs->checksum, d_slots, (uint32_t*)d_sizes, NULL);
    if (rc == ZXC_OK) rc = zxc_hip_block_offsets((uint32_t*)d_sizes, (uint64_t*)d_offs, nb, (uint32_t)bs + 65u, NULL);
    if (rc == ZXC_OK) rc = zxc_mi355x_gather_blocks_device(d_slots, (uint32_t)bs, (const uint32_t*)d_sizes, (const uint64_t*)d_offs, d_out, nb, NULL);
    if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(cs->h_sizes, d_sizes, (size_t)nb * 4); /* (waits for the three launches) */
    size_t csum = 0;
    for (uint32_t i = 0; rc == ZXC_OK && i < nb; i++) {
        /* never trusted as a copy length nor as the place of a trailer (comp_sink above) */
        if (cs->h_sizes[i] > bs + 64 || cs->h_sizes[i] < 8u + (cs->checksum ? 4u : 0u)) rc = ZXC_ERROR_CORRUPT_DATA;
        csum += cs->h_sizes[i];
    }
    uint8_t* land = NULL;
    if (rc == ZXC_OK) {
        if (out->size - out->pos >= csum) {
            land = (uint8_t*)out->dst + out->pos;
        } else {
            rc = ps_host_reserve(&cs->pending, &cs->pending_cap, csum, 0);
            land = cs->pending;
        }
    }
    if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(land, d_out, csum);
    ps_leave_device(cs->dev, prev);
    if (rc != ZXC_OK) return rc;
    if (cs->checksum) { /* the block trailers, folded in stream order (src/lib/zxc_pstream.c:198-203) */
        size_t o = 0;
        for (uint32_t i = 0; i < nb; i++) {
            o += cs->h_sizes[i];
            cs->global_hash = ((cs->global_hash << 1) | (cs->global_hash >> 31)) ^ rd32(land + o - 4);
        }
    }
    if (land == cs->pending) {
        cs->pending_len = csum;
        cs->pending_pos = 0;
    } else {
        out->pos += csum;
        cs->pending_len = cs->pending_pos = 0;
    }
    cs->total_in += total;
    return ZXC_OK;
}

/* -> 1 once pending is empty */
static int cs_drain(zxc_cstream* cs, zxc_outbuf_t* out) {
    const size_t room = out->size - out->pos, have = cs->pending_len - cs->pending_pos;
    const size_t n = room < have ? room : have;
    if (n) {
        memcpy((uint8_t*)out->dst + out->pos, cs->pending + cs->pending_pos, n);
        out->pos += n;
        cs->pending_pos += n;
    }
    return cs->pending_pos == cs->pending_len;
}
static void cs_stage_file_header(zxc_cstream* cs) { /* 16 bytes, src/lib/zxc_common.c:534-558 (no dictionary) */
    uint8_t* h = cs->pending;
    memset(h, 0, ZXC_FILE_HEADER_SIZE);
    wr32(h, MAGIC);
    h[4] = FORMAT_VERSION;
    uint8_t lg = 0;
    while (((size_t)1 << lg) < cs->block_size) lg++;
    h[5] = lg;
    h[6] = cs->checksum ? 0x80 : 0;
    const uint16_t crc = hdr_hash16(h);
    h[14] = (uint8_t)crc;
    h[15] = (uint8_t)(crc >> 8);
    cs->pending_len = ZXC_FILE_HEADER_SIZE;
    cs->pending_pos = 0;
}
static void cs_stage_eof(zxc_cstream* cs) {
    memset(cs->pending, 0, BLK_HDR);
    cs->pending[0] = BLK_EOF;
    cs->pending[7] = hdr_hash8(cs->pending);
    cs->pending_len = BLK_HDR;
    cs->pending_pos = 0;
}
static void cs_stage_footer(zxc_cstream* cs) {
    wr64(cs->pending, cs->total_in);
    wr32(cs->pending + 8, cs->checksum ? cs->global_hash : 0);
    cs->pending_len = ZXC_FILE_FOOTER_SIZE;
    cs->pending_pos = 0;
}

int64_t zxc_cstream_compress(zxc_cstream* cs, zxc_outbuf_t* out, zxc_inbuf_t* in) {
    if (!cs || !out || !in || in->pos > in->size || out->pos > out->size || (in->size > in->pos && !in->src) ||
        (out->size > out->pos && !out->dst) || cs->state == CS_DONE)
        return ZXC_ERROR_NULL_INPUT;
    if (cs->state == CS_ERRORED) return cs->error_code;
    const size_t bs = cs->block_size;
    for (;;) {
        switch (cs->state) {
            case CS_INIT:
                cs_stage_file_header(cs);
                cs->state = CS_DRAIN_HEADER;
                break;
            case CS_DRAIN_HEADER:
            case CS_DRAIN_BLOCK:
                if (!cs_drain(cs, out)) return (int64_t)(cs->pending_len - cs->pending_pos);
                cs->state = CS_ACCUMULATE;
                break;
            case CS_ACCUMULATE: {
                const uint8_t* ip = (const uint8_t*)in->src + in->pos;
                size_t avail = in->size - in->pos;
                if (cs->in_used == 0 && avail < bs) { /* not a block yet: keep it for the next call */
                    if (avail) memcpy(cs->in_block, ip, avail);
                    cs->in_used = avail;
                    in->pos += avail;
                    return 0;
                }
                size_t alen = 0;
                if (cs->in_used) { /* fill the accumulator first */
                    const size_t room = bs - cs->in_used, n = avail < room ? avail : room;
                    if (n) memcpy(cs->in_block + cs->in_used, ip, n);
                    cs->in_used += n;
                    in->pos += n;
                    ip += n;
                    avail -= n;
                    if (cs->in_used < bs) return 0;
                    alen = bs;
                }
                /* the accumulator's block (if any) and every whole block behind it go up together */
                size_t k = avail / bs;
                const size_t kmax = cs->max_blocks - (alen ? 1u : 0u);
                if (k > kmax) k = kmax;
                const int rc = cs_encode(cs, cs->in_block, alen, ip, k * bs, out);
                if (rc != ZXC_OK) return cs_fail(cs, rc);
                cs->in_used = 0;
                in->pos += k * bs;
                cs->state = CS_DRAIN_BLOCK;
                break;
            }
            default: /* CS_DRAIN_LAST / _EOF / _FOOTER: zxc_cstream_end() has begun (src/lib/zxc_pstream.c:499-505) */
                return ZXC_ERROR_NULL_INPUT;
        }
    }
}

int64_t zxc_cstream_end(zxc_cstream* cs, zxc_outbuf_t* out) {
    if (!cs || !out || out->pos > out->size || (out->size > out->pos && !out->dst) || cs->state == CS_DONE) return ZXC_ERROR_NULL_INPUT;
    if (cs->state == CS_ERRORED) return cs->error_code;
    for (;;) {
        switch (cs->state) {
            case CS_INIT: /* end before any input: the header is still owed */
                cs_stage_file_header(cs);
                cs->state = CS_DRAIN_HEADER;
                break;
            case CS_DRAIN_HEADER:
            case CS_DRAIN_BLOCK:
                if (!cs_drain(cs, out)) return (int64_t)(cs->pending_len - cs->pending_pos);
                cs->state = CS_ACCUMULATE;
                break;
            case CS_ACCUMULATE:
                if (cs->in_used > 0) { /* the residual partial block */
                    const int rc = cs_encode(cs, cs->in_block, cs->in_used, NULL, 0, out);
                    if (rc != ZXC_OK) return cs_fail(cs, rc);
                    cs->in_used = 0;
                    cs->state = CS_DRAIN_LAST;
                    break;
                }
                cs_stage_eof(cs);
                cs->state = CS_DRAIN_EOF;
                break;
            case CS_DRAIN_LAST:
                if (!cs_drain(cs, out)) return (int64_t)(cs->pending_len - cs->pending_pos);
                cs_stage_eof(cs);
                cs->state = CS_DRAIN_EOF;
                break;
            case CS_DRAIN_EOF:
                if (!cs_drain(cs, out)) return (int64_t)(cs->pending_len - cs->pending_pos);
                cs_stage_footer(cs);
                cs->state = CS_DRAIN_FOOTER;
                break;
            case CS_DRAIN_FOOTER:
                if (!cs_drain(cs, out)) return (int64_t)(cs->pending_len - cs->pending_pos);
                cs->state = CS_DONE;
                return 0;
            default:
                return cs->state == CS_ERRORED ? cs->error_code : 0;
        }
    }
}

/* ========================================================== decompression */
enum { DS_FILE_HEADER = 0, DS_BLOCK_HEADER, DS_PAYLOAD, DS_FLUSH, DS_EMIT, DS_PEEK_TAIL, DS_SEK, DS_FOOTER, DS_VALIDATE, DS_DONE, DS_ERRORED };

struct zxc_dstream_s {
    int want_verify;       /* opts.checksum_enabled */
    uint32_t block_size;   /* 0 until the file header is parsed */
    int file_ck;
    uint32_t max_blocks;
    size_t window;         /* bytes, fixed at creation */
    uint8_t scratch[32];   /* file header, a block header that straddles calls, the footer */
    size_t scratch_used, scratch_need;
    uint8_t* carry;        /* ONE block frame that straddles calls: header + payload (+ trailer) */
    size_t carry_cap, carry_used, carry_need;
    /* the batch being collected: [carry frame] + frames of the span [span, span + span_len) of the caller's input */
    zxc_dev_job_t* jobs;
    uint32_t n;
    int has_carry;
    const uint8_t* span;
    size_t span_len;
    /* decoded bytes the caller's out had no room for */
    uint8_t* decoded;
    size_t decoded_cap, decoded_size, decoded_pos;
    int next_state;        /* where to go on once a batch is delivered */
    int tail_err;          /* error to raise once a batch is delivered (a failing block, or a bad header found behind the batch) */
    size_t sek_remaining;
    uint64_t total_out;
    uint32_t global_hash;
    int state, error_code;
    int32_t* h_st;
    int dev;
    dbuf_t d_comp, d_jobs, d_out, d_status;
};

static int ds_fail(zxc_dstream* ds, int code) {
    ds->error_code = code;
    ds->state = DS_ERRORED;
    return code;
}

zxc_dstream* zxc_dstream_create(const zxc_decompress_opts_t* opts) {
    if (opts && (opts->dict || opts->dict_size || opts->dict_huf)) return NULL;
    zxc_dstream* ds = (zxc_dstream*)calloc(1, sizeof(*ds));
    if (!ds) return NULL;
    ds->want_verify = opts ? (opts->checksum_enabled != 0) : 0;
    ds->state = DS_FILE_HEADER;
    ds->scratch_need = ZXC_FILE_HEADER_SIZE;
    ds->dev = -1;
    ds->window = PS_WINDOW_BYTES;
    return ds;
}

void zxc_dstream_free(zxc_dstream* ds) {
    if (!ds) return;
    if (ds->dev >= 0) {
        int prev = -1;
        if (ps_enter_device(&ds->dev, &prev) == ZXC_OK) {
            zxc_mi355x_free(ds->d_comp.p); zxc_mi355x_free(ds->d_jobs.p); zxc_mi355x_free(ds->d_out.p); zxc_mi355x_free(ds->d_status.p);
        }
        ps_leave_device(ds->dev, prev);
    }
    free(ds->h_st);
    free(ds->jobs);
    free(ds->decoded);
    free(ds->carry);
    free(ds);
}

int zxc_dstream_finished(const zxc_dstream* ds) { return (ds && ds->state == DS_DONE) ? 1 : 0; }
size_t zxc_dstream_in_size(const zxc_dstream* ds) { return ds ? ds->window : 0; }
size_t zxc_dstream_out_size(const zxc_dstream* ds) {
    if (!ds) return 0;
    return ds->block_size ? (size_t)ds->max_blocks * ds->block_size : ds->window;
}

static int ds_pull_scratch(zxc_dstream* ds, zxc_inbuf_t* in) {
    const size_t want = ds->scratch_need - ds->scratch_used, avail = in->size - in->pos;
    const size_t n = want < avail ? want : avail;
    if (n) {
        memcpy(ds->scratch + ds->scratch_used, (const uint8_t*)in->src + in->pos, n);
        in->pos += n;
        ds->scratch_used += n;
    }
    return ds->scratch_used == ds->scratch_need;
}

/* The launch over the collected batch. Blocks sit back to back in d_out (block i at i * block_size); a batch whose blocks do not
 * all decode to block_size (legal, never written by an encoder of this format: frames glued together from the Block API) runs a
 * second time with one capacity-sized slot per block. What precedes the first failing block is delivered, then the error. */
static int ds_flush(zxc_dstream* ds, zxc_outbuf_t* out, size_t* produced) {
    const uint32_t n = ds->n, bs = ds->block_size;
    ds->decoded_size = ds->decoded_pos = 0;
    if (n == 0) return ZXC_OK;
    int prev = -1;
    int rc = ps_enter_device(&ds->dev, &prev);
    if (rc != ZXC_OK) return rc;
    const size_t carry_area = ds->has_carry ? ((ds->carry_need + 15u) & ~(size_t)15u) : 0;
    const uint32_t slot = (bs + TAIL_PAD + 15u) & ~15u;
    uint8_t* d_comp = (uint8_t*)ps_dev_reserve(&ds->d_comp, carry_area + ds->span_len + 64); /* (+64: the kernel's 16-byte reads) */
    void* d_jobs = ps_dev_reserve(&ds->d_jobs, (size_t)ds->max_blocks * sizeof(zxc_dev_job_t));
    void* d_status = ps_dev_reserve(&ds->d_status, (size_t)ds->max_blocks * sizeof(int32_t));
    uint8_t* d_out = (uint8_t*)ps_dev_reserve(&ds->d_out, (size_t)n * bs + 64);
    if (!d_comp || !d_jobs || !d_status || !d_out) rc = ZXC_ERROR_MEMORY;
    if (rc == ZXC_OK && ds->has_carry) rc = zxc_mi355x_memcpy_h2d(d_comp, ds->carry, ds->carry_need);
    if (rc == ZXC_OK && ds->span_len) rc = zxc_mi355x_memcpy_h2d(d_comp + carry_area, ds->span, ds->span_len);
    const int verify = ds->want_verify && ds->file_ck;
    int32_t* st = ds->h_st;
    uint32_t good = n;
    size_t bytes = 0;
    int regular = 1;
    for (int pass = 0; rc == ZXC_OK && pass < 2; pass++) {
        for (uint32_t i = 0; i < n; i++) {
            ds->jobs[i].out_off = (uint64_t)i * (pass ? slot : bs);
            ds->jobs[i].out_len = pass ? slot : bs;
        }
        rc = zxc_mi355x_memcpy_h2d(d_jobs, ds->jobs, (size_t)n * sizeof(zxc_dev_job_t));
        if (rc == ZXC_OK) rc = zxc_mi355x_decode_blocks_device(d_comp, (const zxc_dev_job_t*)d_jobs, n, d_out, (int32_t*)d_status, bs, verify, NULL);
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(st, d_status, (size_t)n * sizeof(int32_t)); /* (waits for the launch) */
        if (rc != ZXC_OK || pass) break;
        good = n;
        for (uint32_t i = 0; i < n; i++)
            if (st[i] < 0) { good = i; ds->tail_err = st[i]; break; } /* (wins over an error found behind the batch) */
        bytes = 0;
        for (uint32_t i = 0; i < good; i++) {
            if ((uint32_t)st[i] > bs || ((uint32_t)st[i] != bs && i + 1 != good)) regular = 0;
            bytes += (size_t)st[i];
        }
        if (regular) break;
        d_out = (uint8_t*)ps_dev_reserve(&ds->d_out, (size_t)n * slot + 64);
        if (!d_out) rc = ZXC_ERROR_MEMORY;
    }
    if (rc == ZXC_OK && !regular) { /* block by block out of the slots (a status is never trusted as a copy length) */
        bytes = 0;
        for (uint32_t i = 0; i < good && rc == ZXC_OK; i++) {
            if (st[i] < 0 || (uint32_t)st[i] > slot) { rc = st[i] < 0 ? st[i] : ZXC_ERROR_CORRUPT_DATA; break; }
            bytes += (size_t)st[i];
        }
        if (rc == ZXC_OK) rc = ps_host_reserve(&ds->decoded, &ds->decoded_cap, bytes, 0);
        size_t o = 0;
        for (uint32_t i = 0; i < good && rc == ZXC_OK; i++) {
            rc = zxc_mi355x_memcpy_d2h(ds->decoded + o, d_out + (size_t)i * slot, (size_t)st[i]);
            o += (size_t)st[i];
        }
        if (rc == ZXC_OK) ds->decoded_size = bytes;
    } else if (rc == ZXC_OK && bytes) {
        if (out->size - out->pos >= bytes) { /* straight into the caller's buffer */
            rc = zxc_mi355x_memcpy_d2h((uint8_t*)out->dst + out->pos, d_out, bytes);
            if (rc == ZXC_OK) {
                out->pos += bytes;
                *produced += bytes;
                ds->total_out += bytes;
            }
        } else {
            rc = ps_host_reserve(&ds->decoded, &ds->decoded_cap, bytes, 0);
            if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(ds->decoded, d_out, bytes);
            if (rc == ZXC_OK) ds->decoded_size = bytes;
        }
    }
    ps_leave_device(ds->dev, prev);
    ds->n = 0;
    ds->has_carry = 0;
    ds->span = NULL;
    ds->span_len = 0;
    return rc;
}

int64_t zxc_dstream_decompress(zxc_dstream* ds, zxc_outbuf_t* out, zxc_inbuf_t* in) {
    if (!ds || !out || !in || in->pos > in->size || out->pos > out->size || (in->size > in->pos && !in->src) ||
        (out->size > out->pos && !out->dst))
        return ZXC_ERROR_NULL_INPUT;
    if (ds->state == DS_ERRORED) return ds->error_code;
    if (ds->state == DS_DONE) return 0;
    size_t produced = 0;
    for (;;) {
        switch (ds->state) {
            case DS_FILE_HEADER: {
                if (!ds_pull_scratch(ds, in)) return (int64_t)produced;
                uint32_t bs = 0, did = 0;
                int ck = 0;
                const int rc = read_file_header(ds->scratch, ds->scratch_used, &bs, &ck, &did);
                if (rc != ZXC_OK) return ds_fail(ds, rc);
                ds->block_size = bs;
                ds->file_ck = ck;
                ds->max_blocks = (uint32_t)(ds->window / bs < 16 ? 16 : ds->window / bs);
                ds->jobs = (zxc_dev_job_t*)malloc((size_t)ds->max_blocks * sizeof(zxc_dev_job_t));
                ds->h_st = (int32_t*)malloc((size_t)ds->max_blocks * sizeof(int32_t));
                if (!ds->jobs || !ds->h_st) return ds_fail(ds, ZXC_ERROR_MEMORY);
                ds->state = DS_BLOCK_HEADER;
                ds->scratch_used = 0;
                ds->scratch_need = BLK_HDR;
                break;
            }
            case DS_BLOCK_HEADER: {
                const size_t avail = in->size - in->pos;
                const uint8_t* hdr;
                int direct = 0;
                if (ds->scratch_used == 0 && avail >= BLK_HDR) {
                    hdr = (const uint8_t*)in->src + in->pos;
                    direct = 1;
                } else {
                    if (ds->n > 0 && avail < BLK_HDR - ds->scratch_used) { /* the input ends here: the collected blocks first */
                        ds->next_state = DS_BLOCK_HEADER;
                        ds->state = DS_FLUSH;
                        break;
                    }
                    if (!ds_pull_scratch(ds, in)) return (int64_t)produced;
                    hdr = ds->scratch;
                }
                uint8_t type = 0;
                uint32_t csz = 0;
                int herr = read_block_header(hdr, BLK_HDR, &type, &csz);
                if (herr == ZXC_OK && type == BLK_EOF) {
                    if (csz != 0) herr = ZXC_ERROR_BAD_BLOCK_SIZE; /* src/lib/zxc_pstream.c:991 */
                    else {
                        if (direct) in->pos += BLK_HDR;
                        ds->scratch_used = 0;
                        ds->next_state = DS_PEEK_TAIL;
                        ds->state = DS_FLUSH;
                        break;
                    }
                }
                const uint64_t need = (uint64_t)csz + (ds->file_ck ? 4u : 0u);
                if (herr == ZXC_OK && need > zxc_compress_block_bound(ds->block_size)) herr = ZXC_ERROR_BAD_BLOCK_SIZE; /* :1001 */
                if (herr != ZXC_OK) { /* what was collected in front of it decodes (and may fail) first */
                    if (direct) in->pos += BLK_HDR; /* (the reference has pulled the header it rejects) */
                    ds->scratch_used = 0;
                    ds->tail_err = herr;
                    ds->state = DS_FLUSH;
                    break;
                }
                const size_t phys = BLK_HDR + (size_t)need;
                if (direct && avail >= phys) { /* the frame lies whole in the caller's input: it joins the span where it is */
                    if (ds->span == NULL) { ds->span = hdr; ds->span_len = 0; }
                    const size_t carry_area = ds->has_carry ? ((ds->carry_need + 15u) & ~(size_t)15u) : 0;
                    ds->jobs[ds->n].comp_off = carry_area + ds->span_len;
                    ds->jobs[ds->n].comp_size = (uint32_t)phys;
                    ds->n++;
                    ds->span_len += phys;
                    in->pos += phys;
                    if (ds->want_verify && ds->file_ck) ds->global_hash = ((ds->global_hash << 1) | (ds->global_hash >> 31)) ^ rd32(hdr + BLK_HDR + csz);
                    if (ds->n == ds->max_blocks) { ds->next_state = DS_BLOCK_HEADER; ds->state = DS_FLUSH; }
                    break;
                }
                if (ds->n > 0) { /* the frame is cut by the end of the input: the collected blocks first, then come back */
                    ds->next_state = DS_BLOCK_HEADER;
                    ds->state = DS_FLUSH;
                    break;
                }
                if (ps_host_reserve(&ds->carry, &ds->carry_cap, phys, 0) != ZXC_OK) return ds_fail(ds, ZXC_ERROR_MEMORY);
                memcpy(ds->carry, hdr, BLK_HDR);
                if (direct) in->pos += BLK_HDR;
                ds->scratch_used = 0;
                ds->carry_used = BLK_HDR;
                ds->carry_need = phys;
                ds->state = DS_PAYLOAD;
                break;
            }
            case DS_PAYLOAD: {
                const size_t want = ds->carry_need - ds->carry_used, avail = in->size - in->pos;
                const size_t k = want < avail ? want : avail;
                if (k) {
                    memcpy(ds->carry + ds->carry_used, (const uint8_t*)in->src + in->pos, k);
                    in->pos += k;
                    ds->carry_used += k;
                }
                if (ds->carry_used < ds->carry_need) return (int64_t)produced;
                /* (the batch is empty here: a carry only starts behind a flush) */
                ds->jobs[0].comp_off = 0;
                ds->jobs[0].comp_size = (uint32_t)ds->carry_need;
                ds->n = 1;
                ds->has_carry = 1;
                if (ds->want_verify && ds->file_ck) ds->global_hash = ((ds->global_hash << 1) | (ds->global_hash >> 31)) ^ rd32(ds->carry + ds->carry_need - 4);
                ds->state = DS_BLOCK_HEADER;
                break;
            }
            case DS_FLUSH: {
                const int rc = ds_flush(ds, out, &produced);
                if (rc != ZXC_OK) return ds_fail(ds, rc);
                ds->state = DS_EMIT;
                break;
            }
            case DS_EMIT: {
                const size_t room = out->size - out->pos, have = ds->decoded_size - ds->decoded_pos;
                const size_t k = room < have ? room : have;
                if (k) {
                    memcpy((uint8_t*)out->dst + out->pos, ds->decoded + ds->decoded_pos, k);
                    out->pos += k;
                    ds->decoded_pos += k;
                    ds->total_out += k;
                    produced += k;
                }
                if (ds->decoded_pos < ds->decoded_size) return (int64_t)produced;
                if (ds->tail_err) return ds_fail(ds, ds->tail_err);
                ds->state = ds->next_state;
                if (ds->state == DS_PEEK_TAIL) { ds->scratch_used = 0; ds->scratch_need = BLK_HDR; }
                break;
            }
            case DS_PEEK_TAIL: { /* behind the EOF block: a SEK block, or the first 8 bytes of the footer (src/lib/zxc_pstream.c:1122-1137) */
                if (!ds_pull_scratch(ds, in)) return (int64_t)produced;
                uint8_t type = 0;
                uint32_t csz = 0;
                if (read_block_header(ds->scratch, BLK_HDR, &type, &csz) == ZXC_OK && type == BLK_SEK) {
                    ds->sek_remaining = csz;
                    ds->state = DS_SEK;
                } else {
                    ds->scratch_need = ZXC_FILE_FOOTER_SIZE; /* keep the 8, 4 more */
                    ds->state = DS_FOOTER;
                }
                break;
            }
            case DS_SEK: {
                const size_t avail = in->size - in->pos;
                const size_t k = avail < ds->sek_remaining ? avail : ds->sek_remaining;
                in->pos += k;
                ds->sek_remaining -= k;
                if (ds->sek_remaining > 0) return (int64_t)produced;
                ds->scratch_used = 0;
                ds->scratch_need = ZXC_FILE_FOOTER_SIZE;
                ds->state = DS_FOOTER;
                break;
            }
            case DS_FOOTER:
                if (!ds_pull_scratch(ds, in)) return (int64_t)produced;
                ds->state = DS_VALIDATE;
                break;
            case DS_VALIDATE:
                if (rd64(ds->scratch) != ds->total_out) return ds_fail(ds, ZXC_ERROR_CORRUPT_DATA);
                if (ds->want_verify && ds->file_ck && rd32(ds->scratch + 8) != ds->global_hash) return ds_fail(ds, ZXC_ERROR_BAD_CHECKSUM);
                ds->state = DS_DONE;
                return (int64_t)produced;
            default:
                return ds->state == DS_ERRORED ? ds->error_code : (int64_t)produced;
        }
    }
}
