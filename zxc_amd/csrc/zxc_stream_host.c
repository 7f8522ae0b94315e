/* zxc_stream_host.c — FILE* callers of the block path (included at the end of zxc_host.c).
 *
 * Reference: src/lib/zxc_driver.c:627-1030 (zxc_stream_engine_run: reader thread -> ring of
 * per-block jobs -> worker threads -> in-order writer), zxc_stream_compress :1038,
 * zxc_stream_decompress :1066, zxc_stream_get_decompressed_size :1099,
 * zxc_seekable_open_file :1215. Same results, different shape: a batch of blocks is read, handed
 * to ONE device launch (the "workers"), and written out in order; batches bound host and device
 * memory. Error precedence is the reference's: the first failing block in stream order wins and
 * nothing after it is written. */
#include <pthread.h>
#include <stdio.h>
#include <unistd.h>
#include <sys/types.h>
#include <sys/stat.h>
#include "../../include/zxc_stream.h"

/* decoded bytes per launch (default 256 MiB = 4096 blocks of 64 KiB); ZXC_STREAM_BATCH_BYTES overrides (tests) */
static size_t stream_batch_bytes(void) {
    const char* e = getenv("ZXC_STREAM_BATCH_BYTES");
    if (e && *e) {
        const unsigned long long v = strtoull(e, NULL, 10);
        if (v >= 4096ull && v <= ((unsigned long long)8 << 30)) return (size_t)v;
    }
    return (size_t)256 << 20;
}
#define STREAM_BATCH_BYTES (stream_batch_bytes())

int64_t zxc_stream_get_decompressed_size(FILE* f_in) {
    if (!f_in) return ZXC_ERROR_NULL_INPUT;
    const off_t saved = ftello(f_in);
    if (saved < 0) return ZXC_ERROR_IO;
    if (fseeko(f_in, 0, SEEK_END) != 0) return ZXC_ERROR_IO;
    const off_t size = ftello(f_in);
    int64_t ret;
    uint8_t hdr[ZXC_FILE_HEADER_SIZE], foot[ZXC_FILE_FOOTER_SIZE];
    if (size < (off_t)(ZXC_FILE_HEADER_SIZE + ZXC_FILE_FOOTER_SIZE)) ret = ZXC_ERROR_SRC_TOO_SMALL;
    else if (fseeko(f_in, 0, SEEK_SET) != 0 || fread(hdr, 1, sizeof hdr, f_in) != sizeof hdr) ret = ZXC_ERROR_IO;
    else if (rd32(hdr) != MAGIC) ret = ZXC_ERROR_BAD_MAGIC;
    else if (fseeko(f_in, size - (off_t)ZXC_FILE_FOOTER_SIZE, SEEK_SET) != 0 || fread(foot, 1, sizeof foot, f_in) != sizeof foot)
        ret = ZXC_ERROR_IO;
    else ret = (int64_t)rd64(foot);
    if (fseeko(f_in, saved, SEEK_SET) != 0 && ret >= 0) ret = ZXC_ERROR_IO;
    return ret;
}

/* The third stage: the decoded bytes of a batch go to f_out on a writer thread of their own, so that the next batch's upload,
 * launch and download run beside the (slow) file writes — reference: the in-order writer of zxc_stream_engine_run,
 * src/lib/zxc_driver.c:627-1030. One writer at a time, in batch order: a batch's writer starts only when the previous one has
 * finished without an error, so nothing behind a failed write (or a failed block) reaches the file. Two host buffers alternate. */
typedef struct {
    FILE* f_out;
    const uint8_t* host;   /* the batch's slots on the host */
    int32_t* st;           /* its statuses (owned: freed by the writer) */
    uint32_t good, slot;
    int64_t result;        /* bytes written or a negative error */
} stream_write_t;
static void* stream_write_thread(void* p) {
    stream_write_t* w = (stream_write_t*)p;
    int64_t written = 0;
    for (uint32_t i = 0; i < w->good; i++) { /* blocks before the first failure are written, like the reference's writer */
        if (w->st[i] > 0 && fwrite(w->host + (size_t)i * w->slot, 1, (size_t)w->st[i], w->f_out) != (size_t)w->st[i]) {
            written = ZXC_ERROR_IO;
            break;
        }
        written += w->st[i];
    }
    free(w->st);
    w->st = NULL;
    w->result = written;
    return NULL;
}
typedef struct {
    uint8_t* host_out[2];
    size_t host_cap[2];
    int next;              /* the host buffer the next batch takes */
    int live;              /* a writer thread is running */
    pthread_t thread;
    stream_write_t job;
} stream_writer_t;
/* waits for the writer in flight: 0 or its error */
static int64_t stream_writer_join(stream_writer_t* W) {
    if (!W->live) return 0;
    pthread_join(W->thread, NULL);
    W->live = 0;
    return W->job.result < 0 ? W->job.result : 0;
}

/* one decoded batch: launch, check statuses in order, hand the bytes to the writer. Returns the bytes this batch contributes
 * (they are on their way to the file when this returns) or a negative error — this batch's, or an earlier batch's write error. */
static int64_t stream_flush_decode(const uint8_t* comp, size_t comp_used, zxc_dev_job_t* jobs, uint32_t n, uint32_t slot,
                                   uint32_t block_size, int verify, const dict_ref_t* dr, FILE* f_out, stream_writer_t* W) {
    if (n == 0) return 0;
    int32_t* st = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    if (!st) return ZXC_ERROR_MEMORY;
    dev_bufs_t b;
    int rc = run_jobs(comp, comp_used, jobs, n, (size_t)n * slot, block_size, verify, st, &b, dr);
    if (rc != ZXC_OK) { free(st); return rc; }
    int64_t ret = 0;
    uint32_t good = n;
    for (uint32_t i = 0; i < n; i++)
        if (st[i] < 0) { ret = st[i]; good = i; break; }
    const int k = W->next;
    if (f_out && good > 0) {
        const size_t need = (size_t)good * slot;
        if (W->host_cap[k] < need) { /* (buffer k's last writer finished before the previous batch was handed over) */
            uint8_t* nb = (uint8_t*)realloc(W->host_out[k], need);
            if (!nb) { dev_bufs_free(&b); free(st); return ZXC_ERROR_MEMORY; }
            W->host_out[k] = nb;
            W->host_cap[k] = need;
        }
        rc = zxc_mi355x_memcpy_d2h(W->host_out[k], b.d_out, need);
        if (rc != ZXC_OK) { dev_bufs_free(&b); free(st); return rc; }
    }
    dev_bufs_free(&b); /* the arena is free again before the (slow) file writes */
    int64_t planned = 0;
    for (uint32_t i = 0; i < good; i++) planned += st[i];
    /* the previous batch's writes must be through (and fine) before this batch's begin */
    const int64_t prev = stream_writer_join(W);
    if (prev < 0) { free(st); return prev; }
    if (f_out && good > 0) {
        W->job = (stream_write_t){f_out, W->host_out[k], st, good, slot, 0};
        W->next = k ^ 1;
        if (pthread_create(&W->thread, NULL, stream_write_thread, &W->job) == 0) W->live = 1;
        else { /* no thread: write here */
            stream_write_thread(&W->job);
            if (W->job.result < 0) return W->job.result;
        }
    } else {
        free(st);
    }
    return ret < 0 ? ret : planned;
}

/* Three stages in flight: the caller's thread reads batch i+2 from f_in (two batch buffers) while a worker thread uploads,
 * decodes and downloads batch i+1 and the writer thread writes batch i. Batches are decoded one at a time and written in
 * order, so the error precedence above holds. */
typedef struct {
    uint8_t* comp;
    size_t comp_cap, comp_used;
    zxc_dev_job_t* jobs;
    uint32_t n;
} stream_batch_t;
typedef struct {
    const stream_batch_t* b;
    uint32_t slot, block_size;
    int verify, device;
    const dict_ref_t* dr;
    FILE* f_out;
    stream_writer_t* writer;
    int64_t result;
} stream_flush_arg_t;
static void* stream_flush_thread(void* p) {
    stream_flush_arg_t* a = (stream_flush_arg_t*)p;
    if (a->device >= 0) zxc_mi355x_set_device(a->device); /* a new thread starts on device 0 */
    a->result = stream_flush_decode(a->b->comp, a->b->comp_used, a->b->jobs, a->b->n, a->slot, a->block_size, a->verify, a->dr,
                                    a->f_out, a->writer);
    return NULL;
}

int64_t zxc_stream_decompress(FILE* f_in, FILE* f_out, const zxc_decompress_opts_t* opts) {
    if (!f_in) return ZXC_ERROR_NULL_INPUT;
    uint8_t hdr[ZXC_FILE_HEADER_SIZE];
    if (fread(hdr, 1, sizeof hdr, f_in) != sizeof hdr) return ferror(f_in) ? ZXC_ERROR_IO : ZXC_ERROR_SRC_TOO_SMALL;
    uint32_t block_size, dict_id;
    int file_ck;
    const int hrc = read_file_header(hdr, sizeof hdr + ZXC_FILE_FOOTER_SIZE, &block_size, &file_ck, &dict_id);
    if (hrc != ZXC_OK) return hrc;
    const int verify = file_ck && opts && opts->checksum_enabled;
    const uint8_t* dict = opts ? (const uint8_t*)opts->dict : NULL;
    const size_t dict_size = (opts && opts->dict) ? opts->dict_size : 0;
    const uint8_t* dict_huf = (opts && opts->dict) ? (const uint8_t*)opts->dict_huf : NULL;
    if (dict_id != 0) {
        if (!dict || dict_size == 0) return ZXC_ERROR_DICT_REQUIRED;
        if (dict_id_of(dict, dict_size, dict_huf) != dict_id) return ZXC_ERROR_DICT_MISMATCH;
    }
    if (dict_size > ZXC_DICT_SIZE_MAX) return ZXC_ERROR_DICT_TOO_LARGE;
    const dict_ref_t dr = {dict, dict_size, dict_huf};
    if (zxc_mi355x_device_count() <= 0) return ZXC_ERROR_GPU_UNAVAILABLE;

    /* every block gets a slot of the reference's per-block capacity, so irregular frames need no second pass */
    const uint32_t slot = (block_size + TAIL_PAD + 15u) & ~15u;
    uint32_t batch_blocks = (uint32_t)(STREAM_BATCH_BYTES / slot);
    if (batch_blocks < 16u) batch_blocks = 16u;
    stream_batch_t B[2];
    memset(B, 0, sizeof B);
    for (int k = 0; k < 2; k++) {
        B[k].jobs = (zxc_dev_job_t*)malloc((size_t)batch_blocks * sizeof(zxc_dev_job_t));
        B[k].comp_cap = (size_t)1 << 20;
        B[k].comp = (uint8_t*)malloc(B[k].comp_cap);
    }
    stream_writer_t writer;
    memset(&writer, 0, sizeof writer);
    if (!B[0].jobs || !B[0].comp || !B[1].jobs || !B[1].comp) {
        for (int k = 0; k < 2; k++) { free(B[k].jobs); free(B[k].comp); }
        return ZXC_ERROR_MEMORY;
    }
    int cur = 0;              /* the batch being read */
    int inflight = 0;         /* a worker is decoding / writing B[cur ^ 1] */
    pthread_t worker;
    stream_flush_arg_t fa;
    memset(&fa, 0, sizeof fa);
    fa.slot = slot; fa.block_size = block_size; fa.verify = verify; fa.dr = &dr; fa.f_out = f_out;
    fa.writer = &writer; fa.device = zxc_mi355x_get_device();
    uint32_t global_hash = 0;
    int64_t total = 0, ret = 0;
    int saw_eof = 0;
    for (;;) {
        stream_batch_t* b = &B[cur];
        uint8_t bh[BLK_HDR];
        const size_t got = fread(bh, 1, BLK_HDR, f_in);
        uint8_t type = 0;
        uint32_t csz = 0;
        int herr = ZXC_OK;
        if (got != BLK_HDR) herr = ferror(f_in) ? ZXC_ERROR_IO : (got == 0 ? ZXC_ERROR_SRC_TOO_SMALL : ZXC_ERROR_BAD_HEADER);
        else herr = read_block_header(bh, BLK_HDR, &type, &csz);
        if (herr == ZXC_OK && type == BLK_EOF) {
            if (csz != 0) herr = ZXC_ERROR_BAD_HEADER;
            else saw_eof = 1;
        }
        /* no valid block is larger than its raw form plus framing: refuse before buffering a corrupt 4 GiB size */
        if (herr == ZXC_OK && !saw_eof && (uint64_t)csz > (uint64_t)block_size + (block_size >> 3) + 4096u)
            herr = ZXC_ERROR_CORRUPT_DATA;
        const size_t want = herr == ZXC_OK && !saw_eof ? (size_t)csz + (file_ck ? 4u : 0u) : 0;
        if (herr == ZXC_OK && !saw_eof) {
            if (b->comp_used + BLK_HDR + want + 64 > b->comp_cap) {
                size_t nc_cap = b->comp_cap;
                while (b->comp_used + BLK_HDR + want + 64 > nc_cap) nc_cap *= 2;
                uint8_t* nc = (uint8_t*)realloc(b->comp, nc_cap);
                if (!nc) { herr = ZXC_ERROR_MEMORY; }
                else { b->comp = nc; b->comp_cap = nc_cap; }
            }
        }
        if (herr == ZXC_OK && !saw_eof) {
            memcpy(b->comp + b->comp_used, bh, BLK_HDR);
            if (fread(b->comp + b->comp_used + BLK_HDR, 1, want, f_in) != want) herr = ferror(f_in) ? ZXC_ERROR_IO : ZXC_ERROR_SRC_TOO_SMALL;
        }
        if (herr == ZXC_OK && !saw_eof) {
            b->jobs[b->n].comp_off = b->comp_used;
            b->jobs[b->n].comp_size = (uint32_t)(BLK_HDR + want);
            b->jobs[b->n].out_off = (uint64_t)b->n * slot;
            b->jobs[b->n].out_len = slot;
            if (verify) global_hash = ((global_hash << 1) | (global_hash >> 31)) ^ rd32(b->comp + b->comp_used + BLK_HDR + csz);
            b->comp_used += BLK_HDR + want;
            b->n++;
        }
        if (herr != ZXC_OK || saw_eof || b->n == batch_blocks) { /* queued blocks first: an earlier block's error wins */
            if (inflight) { /* the previous batch must be out before this one is decoded */
                pthread_join(worker, NULL);
                inflight = 0;
                if (fa.result < 0) { ret = fa.result; break; }
                total += fa.result;
                if (opts && opts->progress_cb) opts->progress_cb((uint64_t)total, 0, opts->user_data);
            }
            fa.b = b;
            if (herr == ZXC_OK && !saw_eof && pthread_create(&worker, NULL, stream_flush_thread, &fa) == 0) {
                inflight = 1; /* decode + write run while the next batch is read into the other buffer */
                cur ^= 1;
                B[cur].n = 0;
                B[cur].comp_used = 0;
                continue;
            }
            stream_flush_thread(&fa); /* the last batch (or no thread to be had): on this thread */
            if (fa.result < 0) { ret = fa.result; break; }
            total += fa.result;
            b->n = 0;
            b->comp_used = 0;
            if (opts && opts->progress_cb) opts->progress_cb((uint64_t)total, 0, opts->user_data);
            if (herr != ZXC_OK) { ret = herr; break; }
            if (saw_eof) break;
        }
    }
    if (inflight) pthread_join(worker, NULL); /* (only after an error of a later batch: its result no longer matters) */
    { /* the last batch's writes */
        const int64_t wr = stream_writer_join(&writer);
        if (wr < 0 && ret == 0) ret = wr;
    }
    if (ret == 0 && saw_eof) {
        /* what follows the EOF block is (a) the 12-byte footer, or (b) a SEK block then the footer; like the
         * reference (zxc_driver.c:959-992) peek 8 bytes and see whether they are a SEK block header */
        uint8_t foot[ZXC_FILE_FOOTER_SIZE];
        uint8_t t2 = 0;
        uint32_t c2 = 0;
        if (fread(foot, 1, BLK_HDR, f_in) != BLK_HDR) ret = ferror(f_in) ? ZXC_ERROR_IO : ZXC_ERROR_SRC_TOO_SMALL;
        else if (read_block_header(foot, BLK_HDR, &t2, &c2) == ZXC_OK && t2 == BLK_SEK) {
            uint8_t buf[4096];
            uint64_t skip = c2;
            while (ret == 0 && skip > 0) {  /* drain the table */
                const size_t k = skip < sizeof buf ? (size_t)skip : sizeof buf;
                if (fread(buf, 1, k, f_in) != k) ret = ferror(f_in) ? ZXC_ERROR_IO : ZXC_ERROR_SRC_TOO_SMALL;
                skip -= k;
            }
            if (ret == 0 && fread(foot, 1, ZXC_FILE_FOOTER_SIZE, f_in) != ZXC_FILE_FOOTER_SIZE)
                ret = ferror(f_in) ? ZXC_ERROR_IO : ZXC_ERROR_SRC_TOO_SMALL;
        } else if (fread(foot + BLK_HDR, 1, ZXC_FILE_FOOTER_SIZE - BLK_HDR, f_in) != ZXC_FILE_FOOTER_SIZE - BLK_HDR) {
            ret = ferror(f_in) ? ZXC_ERROR_IO : ZXC_ERROR_SRC_TOO_SMALL;
        }
        if (ret == 0) {
            if (rd64(foot) != (uint64_t)total) ret = ZXC_ERROR_CORRUPT_DATA;
            else if (verify && rd32(foot + 8) != global_hash) ret = ZXC_ERROR_BAD_CHECKSUM;
        }
    }
    for (int k = 0; k < 2; k++) { free(B[k].jobs); free(B[k].comp); free(writer.host_out[k]); }
    if (ret < 0) return ret;
    if (f_out && fflush(f_out) != 0) return ZXC_ERROR_IO;
    return total;
}

int64_t zxc_stream_compress(FILE* f_in, FILE* f_out, const zxc_compress_opts_t* opts) {
    if (!f_in) return ZXC_ERROR_NULL_INPUT; /* (f_out NULL: the reference's dry-run mode — everything but the writes, src/lib/zxc_driver.c:1038) */
    const int checksum_enabled = opts ? opts->checksum_enabled : 0;
    const int seekable = opts ? opts->seekable : 0;
    int level = (opts && opts->level > 0) ? opts->level : ZXC_LEVEL_DEFAULT;
    if (level > ZXC_LEVEL_ULTRA) level = ZXC_LEVEL_ULTRA;
    const size_t block_size = (opts && opts->block_size > 0) ? opts->block_size : ZXC_BLOCK_SIZE_DEFAULT;
    if (block_size < ZXC_BLOCK_SIZE_MIN || block_size > ZXC_BLOCK_SIZE_MAX || (block_size & (block_size - 1)))
        return ZXC_ERROR_BAD_BLOCK_SIZE;
    const size_t dict_size = (opts && opts->dict) ? opts->dict_size : 0;
    if (dict_size > ZXC_DICT_SIZE_MAX) return ZXC_ERROR_DICT_TOO_LARGE;
    const uint8_t* dict = dict_size ? (const uint8_t*)opts->dict : NULL;
    const uint8_t* dict_huf = dict_size ? (const uint8_t*)opts->dict_huf : NULL;
    if (zxc_mi355x_device_count() <= 0) return ZXC_ERROR_GPU_UNAVAILABLE;

    uint8_t fh[ZXC_FILE_HEADER_SIZE];
    memset(fh, 0, sizeof fh);
    wr32(fh, MAGIC);
    fh[4] = FORMAT_VERSION;
    uint8_t lg = 0;
    while (((size_t)1 << lg) < block_size) lg++;
    fh[5] = lg;
    fh[6] = checksum_enabled ? 0x80 : 0;
    if (dict_size) { /* HAS_DICTIONARY + dict_id, like zxc_compress (src/lib/zxc_common.c:546-553) */
        fh[6] |= 0x40;
        wr32(fh + 7, dict_id_of(dict, dict_size, dict_huf));
    }
    const uint16_t crc = hdr_hash16(fh);
    fh[14] = (uint8_t)crc;
    fh[15] = (uint8_t)(crc >> 8);
    if (f_out && fwrite(fh, 1, sizeof fh, f_out) != sizeof fh) return ZXC_ERROR_IO;
    int64_t out_total = ZXC_FILE_HEADER_SIZE;

    size_t batch = STREAM_BATCH_BYTES / 4; /* source bytes per launch */
    if (batch < block_size) batch = block_size;
    batch -= batch % block_size;
    const uint32_t bpb = (uint32_t)(batch / block_size);
    const uint32_t stride = zxc_mi355x_encode_slot_stride((uint32_t)block_size);
    uint8_t* h_in = (uint8_t*)malloc(batch);
    uint8_t* h_out = (uint8_t*)malloc((size_t)bpb * (block_size + 64) + 64);
    uint32_t* h_sizes = (uint32_t*)malloc((size_t)bpb * 4);
    uint64_t* h_offs = (uint64_t*)malloc((size_t)bpb * 8);
    void* d_src = zxc_mi355x_malloc(batch + 64);
    void* d_slots = zxc_mi355x_malloc((size_t)bpb * stride);
    void* d_sizes = zxc_mi355x_malloc((size_t)bpb * 4);
    void* d_offs = zxc_mi355x_malloc((size_t)bpb * 8);
    void* d_out = zxc_mi355x_malloc((size_t)bpb * (block_size + 64) + 64);
    void* d_dict = dict_size ? zxc_mi355x_malloc(dict_size + 64) : NULL;
    void* d_work = dict_size ? zxc_mi355x_malloc((size_t)zxc_mi355x_encode_dict_work_size(batch, (uint32_t)block_size, (uint32_t)dict_size)) : NULL;
    uint32_t* all_sizes = NULL; /* seek table entries */
    size_t all_n = 0, all_cap = 0;
    uint64_t src_total = 0;
    uint32_t global_hash = 0;
    int64_t ret = (h_in && h_out && h_sizes && h_offs && d_src && d_slots && d_sizes && d_offs && d_out && (!dict_size || (d_dict && d_work)))
                      ? 0 : ZXC_ERROR_MEMORY;
    if (ret == 0 && dict_size) ret = zxc_mi355x_memcpy_h2d(d_dict, dict, dict_size);
    while (ret == 0) {
        const size_t got = fread(h_in, 1, batch, f_in);
        if (got == 0) {
            if (ferror(f_in)) ret = ZXC_ERROR_IO;
            break;
        }
        const uint32_t nb = (uint32_t)((got + block_size - 1) / block_size);
        int rc = zxc_mi355x_memcpy_h2d(d_src, h_in, got);
        if (rc == ZXC_OK)
            rc = dict_size ? zxc_mi355x_encode_blocks_dict_device(d_src, got, (uint32_t)block_size, level, checksum_enabled, d_dict,
                                                                  (uint32_t)dict_size, d_work, d_slots, (uint32_t*)d_sizes, NULL)
                           : zxc_mi355x_encode_blocks_device(d_src, got, (uint32_t)block_size, level, checksum_enabled, d_slots,
                                                             (uint32_t*)d_sizes, NULL);
        if (rc == ZXC_OK) rc = zxc_mi355x_synchronize(NULL);
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(h_sizes, d_sizes, (size_t)nb * 4);
        uint64_t tot = 0;
        if (rc == ZXC_OK) {
            for (uint32_t i = 0; i < nb; i++) { h_offs[i] = tot; tot += h_sizes[i]; }
            rc = zxc_mi355x_memcpy_h2d(d_offs, h_offs, (size_t)nb * 8);
        }
        if (rc == ZXC_OK)
            rc = zxc_mi355x_gather_blocks_device(d_slots, (uint32_t)block_size, (const uint32_t*)d_sizes, (const uint64_t*)d_offs,
                                                 d_out, nb, NULL);
        if (rc == ZXC_OK) rc = zxc_mi355x_synchronize(NULL);
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(h_out, d_out, (size_t)tot);
        if (rc != ZXC_OK) { ret = rc; break; }
        if (checksum_enabled)
            for (uint32_t i = 0; i < nb; i++)
                global_hash = ((global_hash << 1) | (global_hash >> 31)) ^ rd32(h_out + h_offs[i] + h_sizes[i] - 4);
        if (seekable) {
            if (all_n + nb > all_cap) {
                all_cap = all_cap ? all_cap * 2 : 4096;
                while (all_cap < all_n + nb) all_cap *= 2;
                uint32_t* na = (uint32_t*)realloc(all_sizes, all_cap * 4);
                if (!na) { ret = ZXC_ERROR_MEMORY; break; }
                all_sizes = na;
            }
            memcpy(all_sizes + all_n, h_sizes, (size_t)nb * 4);
            all_n += nb;
        }
        if (f_out && fwrite(h_out, 1, (size_t)tot, f_out) != (size_t)tot) { ret = ZXC_ERROR_IO; break; }
        out_total += (int64_t)tot;
        src_total += got;
        if (opts && opts->progress_cb) opts->progress_cb(src_total, 0, opts->user_data);
        if (got < batch) {
            if (ferror(f_in)) ret = ZXC_ERROR_IO;
            break;
        }
    }
    if (ret == 0) { /* EOF block, optional seek table, footer */
        uint8_t eofb[BLK_HDR];
        memset(eofb, 0, sizeof eofb);
        eofb[0] = BLK_EOF;
        eofb[7] = hdr_hash8(eofb);
        if (f_out && fwrite(eofb, 1, sizeof eofb, f_out) != sizeof eofb) ret = ZXC_ERROR_IO;
        else out_total += BLK_HDR;
        if (ret == 0 && seekable && all_n > 0) {
            if (all_n > 0x3FFFFFFFu) ret = ZXC_ERROR_OVERFLOW;
            else {
                const size_t tsz = zxc_seek_table_size((uint32_t)all_n);
                uint8_t* tbl = (uint8_t*)malloc(tsz);
                if (!tbl) ret = ZXC_ERROR_MEMORY;
                else {
                    const int64_t w = zxc_write_seek_table(tbl, tsz, all_sizes, (uint32_t)all_n);
                    if (w < 0) ret = w;
                    else if (f_out && fwrite(tbl, 1, (size_t)w, f_out) != (size_t)w) ret = ZXC_ERROR_IO;
                    else out_total += w;
                    free(tbl);
                }
            }
        }
        if (ret == 0) {
            uint8_t foot[ZXC_FILE_FOOTER_SIZE];
            wr64(foot, src_total);
            wr32(foot + 8, checksum_enabled ? global_hash : 0);
            if (f_out && (fwrite(foot, 1, sizeof foot, f_out) != sizeof foot || fflush(f_out) != 0)) ret = ZXC_ERROR_IO;
            else out_total += ZXC_FILE_FOOTER_SIZE;
        }
    }
    free(h_in); free(h_out); free(h_sizes); free(h_offs); free(all_sizes);
    zxc_mi355x_free(d_src); zxc_mi355x_free(d_slots); zxc_mi355x_free(d_sizes); zxc_mi355x_free(d_offs); zxc_mi355x_free(d_out);
    zxc_mi355x_free(d_dict); zxc_mi355x_free(d_work);
    return ret < 0 ? ret : out_total;
}

/* positional reads on the FILE*'s descriptor: thread-safe, leaves the stream position alone */
static int64_t file_read_at(void* ctx, void* dst, size_t len, uint64_t offset) {
    const int fd = (int)(intptr_t)ctx;
    size_t done = 0;
    while (done < len) {
        const ssize_t r = pread(fd, (uint8_t*)dst + done, len - done, (off_t)(offset + done));
        if (r < 0) return ZXC_ERROR_IO;
        if (r == 0) break;
        done += (size_t)r;
    }
    return (int64_t)done;
}

zxc_seekable* zxc_seekable_open_file(FILE* f) {
    if (!f) return NULL;
    const int fd = fileno(f);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return NULL;
    zxc_reader_t r;
    r.read_at = file_read_at;
    r.ctx = (void*)(intptr_t)fd;
    r.size = (uint64_t)sb.st_size;
    return zxc_seekable_open_reader(&r);
}
