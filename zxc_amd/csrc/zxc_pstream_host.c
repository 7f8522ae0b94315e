/* zxc_pstream_host.c — push streaming (included at the end of zxc_host.c, like zxc_stream_host.c; a .c so that
 * bench.kernel_sources_hash(), which covers the device sources *.hip / *.inc / *.h of this directory, stays a property of the
 * kernels).
 *
 * Reference: include/zxc_pstream.h:82-292, src/lib/zxc_pstream.c (cstream states :69-79, zxc_cstream_compress :446-508,
 * zxc_cstream_end :525-591; dstream states :634-647, zxc_dstream_decompress :1046-1176). Same entry points, return values and
 * error codes; the difference is the unit of work. The reference calls its block codec once per full block on the calling
 * thread. Here one call hands EVERY block its input completes to the device, through the piece pipeline of zxc_host.c (pieces of
 * blocks on the streams of the staging arenas: upload of piece i+1 beside the launch of piece i beside the download of piece
 * i-1; a batch of one piece runs in series on the calling thread):
 *   compress:  [accumulator block (when the previous calls left a partial one and this input fills it)] + all whole blocks of
 *              the caller's input, uploaded from where they are, encoded, compacted on the device and copied back — straight
 *              into the caller's out when it has the room, else into the pending buffer that later calls drain;
 *   decompress: the block frames that lie whole inside the caller's input are uploaded from where they are (no staging copy),
 *              in front of them the one frame that straddled the previous call (the carry buffer); statuses checked in stream
 *              order (first failing block wins, what precedes it is still delivered), output copied straight into out when it
 *              fits, else staged and drained.
 * A batch never outlives the call that collected it, so "the call returned 0" means what it means in the reference. A context
 * owns no device memory: the pieces run in the library's staging arenas. */
#include "../../include/zxc_pstream.h"

/* The window: how much of a call's input becomes one batch when the output cannot go straight into the caller's out (what the
 * pending / staging buffer of a context can grow to), and the chunk size the *_in_size() calls suggest: 128 MiB = 2 048 blocks of
 * 64 KiB, one full round of the encoder's workgroups (256 CUs x 8; 32 MiB per call: 8.5 GB/s of source, 128 MiB: 18 GB/s in
 * series, profiles/r5p_pstream_bench.log). A call whose out has room for everything batches ALL its input.
 * ZXC_MI355X_PSTREAM_WINDOW_MIB = 1 .. 1024 overrides it when a context is created. */
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
static int ps_host_reserve(uint8_t** p, size_t* cap, size_t need, int keep) {
    if (*cap >= need) return ZXC_OK;
    uint8_t* nb = keep ? (uint8_t*)realloc(*p, need) : (free(*p), (uint8_t*)malloc(need));
    if (!nb) { if (!keep) { *p = NULL; *cap = 0; } return ZXC_ERROR_MEMORY; }
    *p = nb;
    *cap = need;
    return ZXC_OK;
}
/* ============================================================ compression */
enum { CS_INIT = 0, CS_DRAIN_HEADER, CS_ACCUMULATE, CS_DRAIN_BLOCK, CS_DRAIN_LAST, CS_DRAIN_EOF, CS_DRAIN_FOOTER, CS_DONE, CS_ERRORED };

struct zxc_cstream_s {
    int level, checksum;
    size_t block_size;
    uint32_t max_blocks;  /* blocks of one window */
    uint8_t* in_block;    /* one block: input that does not yet make a whole block */
    size_t in_used;
    uint8_t* pending;     /* output the caller has not drained yet */
    size_t pending_cap, pending_len, pending_pos;
    uint64_t total_in;
    uint32_t global_hash;
    int state, error_code;
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
    cs->in_block = (uint8_t*)malloc(bs);
    cs->pending_cap = 64;
    cs->pending = (uint8_t*)malloc(cs->pending_cap);
    if (!cs->in_block || !cs->pending) { zxc_cstream_free(cs); return NULL; }
    cs->state = CS_INIT;
    return cs;
}

void zxc_cstream_free(zxc_cstream* cs) {
    if (!cs) return;
    free(cs->pending);
    free(cs->in_block);
    free(cs);
}

size_t zxc_cstream_in_size(const zxc_cstream* cs) { return cs ? (size_t)cs->max_blocks * cs->block_size : 0; }
size_t zxc_cstream_out_size(const zxc_cstream* cs) {
    return cs ? (size_t)cs->max_blocks * (size_t)zxc_compress_block_bound(cs->block_size) : 0;
}

/* The blocks of [a[0..alen) | b[0..blen)] (a: the accumulator — one block, or the stream's last partial one; b: whole blocks in the
 * caller's input) through the encode pipeline of zxc_compress (comp_enqueue / comp_sink above): the accumulator's block is a piece
 * of its own, the blocks of b follow in pieces. Their frames land back to back in out (when it has room for their bound) or in
 * pending. */
typedef struct {
    const uint8_t *a, *b;
    size_t alen, blen, block_size;
    uint32_t next, nb;
} cs_src_t;
static int cs_source(void* ctx, uint32_t max_blocks, pipe_piece_t* p) {
    cs_src_t* c = (cs_src_t*)ctx;
    const uint32_t f = c->next;
    if (f >= c->nb) return 0;
    if (c->alen && f == 0) {
        p->h_comp = c->a;
        p->comp_bytes = c->alen;
        p->n = 1;
    } else {
        const uint32_t fb = f - (c->alen ? 1u : 0u), n = c->nb - f < max_blocks ? c->nb - f : max_blocks;
        const size_t o = (size_t)fb * c->block_size;
        p->h_comp = c->b + o;
        p->comp_bytes = c->blen - o < (size_t)n * c->block_size ? c->blen - o : (size_t)n * c->block_size;
        p->n = n;
    }
    p->out_bytes = (size_t)p->n * (c->block_size + 64);
    p->cookie[0] = f;
    c->next = f + p->n;
    return 1;
}
static int cs_encode(zxc_cstream* cs, const uint8_t* a, size_t alen, const uint8_t* b, size_t blen, zxc_outbuf_t* out) {
    const size_t bs = cs->block_size;
    const uint32_t nb = (alen ? 1u : 0u) + (uint32_t)((blen + bs - 1) / bs);
    if (nb == 0) return ZXC_OK;
    const size_t bound = (size_t)nb * (bs + 80); /* (a block's frame is at most bs + 64: comp_sink) */
    uint8_t* land;
    if (out->size - out->pos >= bound) {
        land = (uint8_t*)out->dst + out->pos;
    } else {
        if (ps_host_reserve(&cs->pending, &cs->pending_cap, bound, 0) != ZXC_OK) return ZXC_ERROR_MEMORY;
        land = cs->pending;
    }
    size_t piece_bytes = COMP_PIECE_BYTES;
    { const char* e = getenv("ZXC_MI355X_FRAME_BATCH_MIB"); if (e && atoi(e) >= 1 && atoi(e) <= 1024) piece_bytes = (size_t)atoi(e) << 20; }
    const uint32_t piece_blocks = (uint32_t)(piece_bytes / bs > 16 ? piece_bytes / bs : 16);
    cs_src_t src = {a, b, alen, blen, bs, 0, nb};
    comp_enq_t ce = {cs->level, cs->checksum};
    comp_sink_t ck;
    memset(&ck, 0, sizeof ck);
    ck.dst = land;
    ck.dst_capacity = bound;
    ck.block_size = bs;
    ck.global_hash = cs->global_hash;
    ck.checksum = cs->checksum;
    ck.sizes = (uint32_t*)malloc((size_t)nb * sizeof(uint32_t));
    ck.offs = (uint64_t*)malloc((size_t)piece_blocks * sizeof(uint64_t));
    int rc = (ck.sizes && ck.offs) ? 0 : ZXC_ERROR_MEMORY;
    if (rc == 0)
        rc = pipe_run_ex(cs_source, &src, comp_sink, &ck, comp_enqueue, &ce, (uint32_t)bs, 0, NULL, alen + blen, (uint64_t)nb * bs,
                         (size_t)piece_blocks * bs, (size_t)16 << 20, 0, PIPE_SLOTS);
    free(ck.sizes);
    free(ck.offs);
    if (rc != 0) return rc < 0 ? rc : ZXC_ERROR_CORRUPT_DATA;
    cs->global_hash = ck.global_hash; /* (the block trailers, folded in stream order: src/lib/zxc_pstream.c:198-203) */
    if (land == cs->pending) {
        cs->pending_len = ck.op;
        cs->pending_pos = 0;
    } else {
        out->pos += ck.op;
        cs->pending_len = cs->pending_pos = 0;
    }
    cs->total_in += alen + blen;
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
                /* everything at once when the frames can land in the caller's out; else a window at a time (pending stays bounded) */
                if (k > kmax && out->size - out->pos < (k + 1) * (bs + 80)) k = kmax;
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
    uint32_t max_blocks;   /* blocks of one window */
    size_t window;         /* bytes, fixed at creation */
    uint8_t scratch[32];   /* file header, a block header that straddles calls, the footer */
    size_t scratch_used, scratch_need;
    uint8_t* carry;        /* ONE block frame that straddles calls: header + payload (+ trailer) */
    size_t carry_cap, carry_used, carry_need;
    /* the batch being collected: [carry frame] + frames of the span [span, span + span_len) of the caller's input; a job's
     * comp_off is its place in that layout (the carry frame at 0, the span behind it at carry_area) */
    zxc_dev_job_t* jobs;
    uint32_t n, jobs_cap;
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
    ds->window = PS_WINDOW_BYTES;
    return ds;
}

void zxc_dstream_free(zxc_dstream* ds) {
    if (!ds) return;
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

/* The collected batch through the decode pipeline (pipe_run above): the carry frame is a piece of its own, the frames of the span
 * follow in pieces, each uploaded from the caller's input where it lies; a piece's blocks sit back to back in its output (block j
 * at j * block_size) and go to `land` in stream order. What precedes the first failing block is delivered, then its code. */
#define DS_STOP_AT_FAILURE 2 /* a sink verdict like PIPE_IRREGULAR */
static size_t ds_carry_area(const zxc_dstream* ds) { return ds->has_carry ? ((ds->carry_need + 15u) & ~(size_t)15u) : 0; }
typedef struct {
    zxc_dstream* ds;
    uint32_t next;
} ds_src_t;
static int ds_source(void* ctx, uint32_t max_blocks, pipe_piece_t* p) {
    ds_src_t* c = (ds_src_t*)ctx;
    const zxc_dstream* ds = c->ds;
    const uint32_t i0 = c->next, bs = ds->block_size;
    if (i0 >= ds->n) return 0;
    uint32_t cnt;
    if (ds->has_carry && i0 == 0) {
        cnt = 1;
        p->h_comp = ds->carry;
        p->comp_bytes = ds->carry_need;
        p->jobs[0].comp_off = 0;
        p->jobs[0].comp_size = ds->jobs[0].comp_size;
    } else {
        cnt = ds->n - i0 < max_blocks ? ds->n - i0 : max_blocks;
        const uint64_t lo = ds->jobs[i0].comp_off, hi = ds->jobs[i0 + cnt - 1].comp_off + ds->jobs[i0 + cnt - 1].comp_size;
        p->h_comp = ds->span + (size_t)(lo - ds_carry_area(ds));
        p->comp_bytes = (size_t)(hi - lo);
        for (uint32_t j = 0; j < cnt; j++) {
            p->jobs[j].comp_off = ds->jobs[i0 + j].comp_off - lo;
            p->jobs[j].comp_size = ds->jobs[i0 + j].comp_size;
        }
    }
    for (uint32_t j = 0; j < cnt; j++) {
        p->jobs[j].out_off = (uint64_t)j * bs;
        p->jobs[j].out_len = bs;
    }
    p->n = cnt;
    p->out_bytes = (size_t)cnt * bs;
    p->cookie[0] = i0;
    c->next = i0 + cnt;
    return 1;
}
typedef struct {
    zxc_dstream* ds;
    uint8_t* land;
    size_t cap, landed;
    uint32_t done; /* blocks delivered (or found failing) so far, in stream order */
} ds_sink_t;
static int ds_sink(void* ctx, const pipe_piece_t* p, const int32_t* st, const dev_bufs_t* b) {
    ds_sink_t* k = (ds_sink_t*)ctx;
    zxc_dstream* ds = k->ds;
    const uint32_t i0 = (uint32_t)p->cookie[0], n = p->n, bs = ds->block_size;
    uint32_t good = n;
    for (uint32_t i = 0; i < n; i++)
        if (st[i] < 0) { good = i; break; }
    /* back to back in the piece's output only if every block but the last delivered one decodes to block_size (a short block in
     * the middle is legal, never written by an encoder of this format: the caller takes the piece slot by slot) */
    size_t bytes = 0;
    for (uint32_t i = 0; i < good; i++) {
        if ((uint32_t)st[i] > bs || ((uint32_t)st[i] != bs && i + 1 != good)) { k->done = i0; return PIPE_IRREGULAR; }
        bytes += (size_t)st[i];
    }
    if (bytes > k->cap - k->landed) return ZXC_ERROR_OVERFLOW; /* (cannot happen: cap = n * block_size) */
    const int rc = zxc_mi355x_memcpy_d2h(k->land + k->landed, b->d_out, bytes);
    if (rc != ZXC_OK) return rc;
    k->landed += bytes;
    k->done = i0 + good;
    if (good < n) { ds->tail_err = st[good]; return DS_STOP_AT_FAILURE; } /* (wins over an error found behind the batch) */
    return 0;
}
/* blocks [j0, n) one capacity-sized slot each, a window at a time, appended to the staging buffer block by block (in series: this
 * path exists for correctness) */
static int ds_flush_slots(zxc_dstream* ds, uint32_t j0) {
    const uint32_t bs = ds->block_size, slot = (bs + TAIL_PAD + 15u) & ~15u;
    const uint32_t win = ds->max_blocks;
    zxc_dev_job_t* jobs = (zxc_dev_job_t*)malloc((size_t)win * sizeof(zxc_dev_job_t));
    int32_t* st = (int32_t*)malloc((size_t)win * sizeof(int32_t));
    int rc = (jobs && st) ? ZXC_OK : ZXC_ERROR_MEMORY;
    const int verify = ds->want_verify && ds->file_ck;
    /* (a failure flag of its own: DS_BLOCK_HEADER may have set tail_err for a bad header BEHIND the collected batch, and what was
     *  collected in front of it decodes first — a block that fails here wins over that header error, as in ds_sink: ADVICE r5) */
    int block_err = 0;
    while (rc == ZXC_OK && j0 < ds->n && !block_err) {
        uint32_t cnt;
        const uint8_t* h_comp;
        size_t comp_bytes;
        if (ds->has_carry && j0 == 0) {
            cnt = 1;
            h_comp = ds->carry;
            comp_bytes = ds->carry_need;
            jobs[0].comp_off = 0;
        } else {
            cnt = ds->n - j0 < win ? ds->n - j0 : win;
            const uint64_t lo = ds->jobs[j0].comp_off, hi = ds->jobs[j0 + cnt - 1].comp_off + ds->jobs[j0 + cnt - 1].comp_size;
            h_comp = ds->span + (size_t)(lo - ds_carry_area(ds));
            comp_bytes = (size_t)(hi - lo);
            for (uint32_t j = 0; j < cnt; j++) jobs[j].comp_off = ds->jobs[j0 + j].comp_off - lo;
        }
        for (uint32_t j = 0; j < cnt; j++) {
            jobs[j].comp_size = ds->jobs[j0 + j].comp_size;
            jobs[j].out_off = (uint64_t)j * slot;
            jobs[j].out_len = slot;
        }
        dev_bufs_t b;
        rc = run_jobs_on(h_comp, comp_bytes, jobs, cnt, (size_t)cnt * slot, bs, 0u, verify, st, &b, NULL, 0, NULL);
        if (rc != ZXC_OK) break;
        for (uint32_t j = 0; j < cnt && rc == ZXC_OK; j++) {
            if (st[j] < 0) { block_err = st[j]; break; }
            if ((uint32_t)st[j] > slot) { rc = ZXC_ERROR_CORRUPT_DATA; break; } /* (a status is never trusted as a copy length) */
            rc = ps_host_reserve(&ds->decoded, &ds->decoded_cap, ds->decoded_size + (size_t)st[j], 1);
            if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(ds->decoded + ds->decoded_size, (const uint8_t*)b.d_out + (size_t)j * slot, (size_t)st[j]);
            if (rc == ZXC_OK) ds->decoded_size += (size_t)st[j];
        }
        dev_bufs_free(&b);
        j0 += cnt;
    }
    if (block_err) ds->tail_err = block_err;
    free(jobs);
    free(st);
    return rc;
}
static int ds_flush(zxc_dstream* ds, zxc_outbuf_t* out, size_t* produced) {
    const uint32_t n = ds->n, bs = ds->block_size;
    ds->decoded_size = ds->decoded_pos = 0;
    if (n == 0) return ZXC_OK;
    const size_t most = (size_t)n * bs;
    ds_sink_t k;
    memset(&k, 0, sizeof k);
    k.ds = ds;
    k.cap = most;
    const int direct = out->size - out->pos >= most;
    if (direct) k.land = (uint8_t*)out->dst + out->pos; /* straight into the caller's buffer */
    else {
        if (ps_host_reserve(&ds->decoded, &ds->decoded_cap, most, 0) != ZXC_OK) return ZXC_ERROR_MEMORY;
        k.land = ds->decoded;
    }
    ds_src_t src = {ds, 0};
    int rc = pipe_run(ds_source, &src, ds_sink, &k, bs, ds->want_verify && ds->file_ck, NULL, ds_carry_area(ds) + ds->span_len, most, 0, 0,
                      PIPE_DECODE_SLOTS);
    if (rc >= 0) {
        if (direct) {
            out->pos += k.landed;
            *produced += k.landed;
            ds->total_out += k.landed;
        } else {
            ds->decoded_size = k.landed;
        }
        rc = rc == PIPE_IRREGULAR ? ds_flush_slots(ds, k.done) : ZXC_OK;
    }
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
                ds->jobs_cap = 256;
                ds->jobs = (zxc_dev_job_t*)malloc((size_t)ds->jobs_cap * sizeof(zxc_dev_job_t));
                if (!ds->jobs) return ds_fail(ds, ZXC_ERROR_MEMORY);
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
                    if (ds->n == ds->jobs_cap) {
                        zxc_dev_job_t* nj = (zxc_dev_job_t*)realloc(ds->jobs, (size_t)ds->jobs_cap * 2 * sizeof(zxc_dev_job_t));
                        if (!nj) return ds_fail(ds, ZXC_ERROR_MEMORY);
                        ds->jobs = nj;
                        ds->jobs_cap *= 2;
                    }
                    if (ds->span == NULL) { ds->span = hdr; ds->span_len = 0; }
                    ds->jobs[ds->n].comp_off = ds_carry_area(ds) + ds->span_len;
                    ds->jobs[ds->n].comp_size = (uint32_t)phys;
                    ds->n++;
                    ds->span_len += phys;
                    in->pos += phys;
                    if (ds->want_verify && ds->file_ck) ds->global_hash = ((ds->global_hash << 1) | (ds->global_hash >> 31)) ^ rd32(hdr + BLK_HDR + csz);
                    /* a window is a batch — unless the caller's out has room for more: then everything the input holds (what does
                     * not land in out has to fit the staging buffer) */
                    if (ds->n >= ds->max_blocks && (uint64_t)(ds->n + 1) * ds->block_size > (uint64_t)(out->size - out->pos)) {
                        ds->next_state = DS_BLOCK_HEADER;
                        ds->state = DS_FLUSH;
                    }
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
