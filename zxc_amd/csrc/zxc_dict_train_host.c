/* zxc_dict_train_host.c — dictionary training (included at the end of zxc_host.c; a .c for the reason zxc_pstream_host.c gives).
 *
 * Reference: include/zxc_dict.h:130-191, src/lib/zxc_dict.c:207-638.
 *   zxc_train_dict      k-gram coverage selection over the concatenated samples (:309-477). Pure host arithmetic on the samples,
 *                       restated here step for step (same hash, same sampling strides, same heap order for ties, same placement):
 *                       the dictionary is BYTE FOR BYTE the reference's (tests/test_dict_train.py, against oracle/_ref).
 *   zxc_train_dict_huf  the shared literal table: the reference compresses 4 KiB slices of the samples against the dictionary and
 *                       histograms the literals its parser leaves (:490-588). That is a caller of the block encoder: here the
 *                       slices go to the device's dictionary encoder in ONE launch (zxc_mi355x_encode_blocks_dict_device, level 5:
 *                       deep chains, literals stored raw or run-length coded), the literal sections of the blocks that come back
 *                       are histogrammed on the host, and a length-limited (8-bit, like ZXC_HUF_MAX_CODE_LEN_DENSITY) Huffman
 *                       code is built by package-merge. The table is a valid shared table for either library; it is not the
 *                       reference's bytes (another parser leaves other literals) — the reference's optional "nudge" pass
 *                       (src/lib/zxc_huffman.c:803) is not applied.
 *   zxc_dict_train      both, then zxc_dict_save (:600-638). */

#define DT_KGRAM 5u                 /* ZXC_DICT_KGRAM_LEN = ZXC_LZ_MIN_MATCH_LEN, src/lib/zxc_internal.h:393, :517 */
#define DT_HASH_BITS 16             /* :395 */
#define DT_MAX_SEGMENTS (1u << 16)  /* :397 */
#define DT_SAMPLE_TARGET (1u << 19) /* :401 */
#define DT_SEG_MAX_LEN 4096u        /* src/lib/zxc_dict.c:381 */
#define DT_HUF_SLICE 4096u          /* ZXC_DICT_HUF_TRAIN_BLOCK, src/lib/zxc_internal.h:405 */
#define DT_HUF_BUDGET (8u << 20)    /* ZXC_DICT_HUF_SAMPLE_BUDGET, :408 */
#define DT_HUF_MAX_LEN 8            /* ZXC_HUF_MAX_CODE_LEN_DENSITY, :602 */

static uint32_t dt_hash(const uint8_t* p) { /* src/lib/zxc_dict.c:231-235 */
    return ((rd32(p) ^ (uint32_t)p[4]) * 0x2D35182Du) >> (32 - DT_HASH_BITS);
}

typedef struct { uint32_t off; uint16_t len; uint32_t score; } dt_seg_t;

/* descending by score through a min-heap, exactly the reference's order for equal scores (src/lib/zxc_dict.c:256-299: dictionaries
 * are reproducible because the order is fixed there, so it is fixed the same way here) */
static void dt_sink(dt_seg_t* a, size_t at, size_t n) {
    for (;;) {
        size_t c = 2 * at + 1;
        if (c >= n) return;
        if (c + 1 < n && a[c + 1].score < a[c].score) c++;
        if (a[at].score <= a[c].score) return;
        const dt_seg_t t = a[at];
        a[at] = a[c];
        a[c] = t;
        at = c;
    }
}
static void dt_sort_desc(dt_seg_t* a, size_t n) {
    if (n < 2) return;
    for (size_t i = n / 2; i > 0; i--) dt_sink(a, i - 1, n);
    for (size_t end = n - 1; end > 0; end--) {
        const dt_seg_t t = a[0];
        a[0] = a[end];
        a[end] = t;
        dt_sink(a, 0, end);
    }
}

static size_t dt_tail(uint8_t* out, size_t cap, const uint8_t* corpus, size_t n) { /* no pattern worth picking: the corpus' tail */
    const size_t k = n < cap ? n : cap;
    memcpy(out, corpus + n - k, k);
    return k;
}

int64_t zxc_train_dict(const void* const* samples, const size_t* sample_sizes, const size_t n_samples, void* dict_buf,
                       const size_t dict_capacity) {
    if (!samples || !sample_sizes || n_samples == 0 || !dict_buf || dict_capacity == 0) return ZXC_ERROR_NULL_INPUT;
    if (dict_capacity > ZXC_DICT_SIZE_MAX) return ZXC_ERROR_DICT_TOO_LARGE;
    size_t n = 0;
    for (size_t i = 0; i < n_samples; i++) n += sample_sizes[i];
    if (n < DT_KGRAM) return ZXC_ERROR_SRC_TOO_SMALL;
    uint8_t* corpus = (uint8_t*)malloc(n);
    uint16_t* freq = (uint16_t*)calloc((size_t)1 << DT_HASH_BITS, sizeof(uint16_t));
    const size_t seg_cap = n / DT_KGRAM < DT_MAX_SEGMENTS ? n / DT_KGRAM : DT_MAX_SEGMENTS;
    dt_seg_t* segs = (dt_seg_t*)malloc((seg_cap ? seg_cap : 1) * sizeof(dt_seg_t));
    if (!corpus || !freq || !segs) { free(corpus); free(freq); free(segs); return ZXC_ERROR_MEMORY; }
    for (size_t i = 0, at = 0; i < n_samples; i++) {
        if (sample_sizes[i]) memcpy(corpus + at, samples[i], sample_sizes[i]);
        at += sample_sizes[i];
    }
    /* k-gram frequencies, sampled so that the 16-bit counters do not saturate (:343-350) */
    const size_t kgrams = n - DT_KGRAM + 1;
    const size_t fstep = kgrams / DT_SAMPLE_TARGET ? kgrams / DT_SAMPLE_TARGET : 1;
    for (size_t i = 0; i < kgrams; i += fstep) {
        uint16_t* f = &freq[dt_hash(corpus + i)];
        if (*f < UINT16_MAX) (*f)++;
    }
    /* candidate segments, their starts spread over the whole corpus; score = summed frequency of the k-grams (:355-392) */
    size_t step = DT_KGRAM;
    if (seg_cap > 0 && n / seg_cap > step) step = n / seg_cap;
    size_t n_segs = 0;
    for (size_t i = 0; i + DT_KGRAM <= n && n_segs < seg_cap; i += step) {
        const uint16_t f = freq[dt_hash(corpus + i)];
        if (f < 2) continue;
        uint32_t cover = f;
        size_t end = i + DT_KGRAM;
        while (end + DT_KGRAM <= n && end - i < DT_SEG_MAX_LEN) {
            const uint16_t nf = freq[dt_hash(corpus + end)];
            if (nf < 2) break;
            cover += nf;
            end += DT_KGRAM;
        }
        segs[n_segs].off = (uint32_t)i;
        segs[n_segs].len = (uint16_t)(end - i);
        segs[n_segs].score = cover;
        n_segs++;
    }
    uint8_t* out = (uint8_t*)dict_buf;
    size_t filled = 0;
    if (n_segs == 0) {
        filled = dt_tail(out, dict_capacity, corpus, n);
    } else {
        /* greedy by coverage; a pick zeroes its k-grams, a segment more than half covered by earlier picks is skipped (:404-439) */
        dt_sort_desc(segs, n_segs);
        size_t picked = 0, total = 0;
        for (size_t i = 0; i < n_segs && total < dict_capacity; i++) {
            const size_t s0 = segs[i].off, s1 = s0 + segs[i].len;
            uint32_t now = 0;
            for (size_t p = s0; p + DT_KGRAM <= s1; p += DT_KGRAM) now += freq[dt_hash(corpus + p)];
            if (now * 2 < segs[i].score) continue;
            size_t take = segs[i].len;
            if (take > dict_capacity - total) take = dict_capacity - total;
            for (size_t p = s0; p + DT_KGRAM <= s1; p += DT_KGRAM) freq[dt_hash(corpus + p)] = 0;
            segs[picked].off = (uint32_t)s0;
            segs[picked].len = (uint16_t)take;
            picked++;
            total += take;
        }
        /* the best pick last: nearest to the data, smallest offsets (:443-452) */
        for (size_t i = picked; i > 0; i--) {
            memcpy(out + filled, corpus + segs[i - 1].off, segs[i - 1].len);
            filled += segs[i - 1].len;
        }
        if (filled == 0) filled = dt_tail(out, dict_capacity, corpus, n);
    }
    free(segs);
    free(freq);
    free(corpus);
    return (int64_t)filled;
}

/* Length-limited Huffman code lengths by package-merge (Larmore & Hirschberg): DT_HUF_MAX_LEN rounds over the sorted weights; an
 * item is a leaf or a package of two items of the round before; a symbol's length is the number of chosen items it is in. n <= 256
 * symbols and 8 rounds: every item simply carries its symbols' multiplicities. */
typedef struct { uint64_t w; uint8_t cnt[256]; } pm_item_t;
static int pm_cmp_leaf(const void* a, const void* b) {
    const uint64_t wa = ((const uint64_t*)a)[0], wb = ((const uint64_t*)b)[0];
    if (wa != wb) return wa < wb ? -1 : 1;
    return ((const uint64_t*)a)[1] < ((const uint64_t*)b)[1] ? -1 : 1; /* (ties by symbol: deterministic) */
}
static int huf_lengths_limited(const uint32_t* freq, uint8_t* len_out) {
    memset(len_out, 0, 256);
    uint64_t leaf[256][2];
    int n = 0;
    for (int s = 0; s < 256; s++)
        if (freq[s]) { leaf[n][0] = freq[s]; leaf[n][1] = (uint64_t)s; n++; }
    if (n == 0) return ZXC_ERROR_CORRUPT_DATA;
    if (n == 1) { len_out[leaf[0][1]] = 1; return ZXC_OK; } /* (a degenerate code of one bit, like the reference's) */
    qsort(leaf, (size_t)n, sizeof leaf[0], pm_cmp_leaf);
    pm_item_t* prev = (pm_item_t*)calloc((size_t)2 * n, sizeof(pm_item_t));
    pm_item_t* cur = (pm_item_t*)calloc((size_t)2 * n, sizeof(pm_item_t));
    if (!prev || !cur) { free(prev); free(cur); return ZXC_ERROR_MEMORY; }
    int n_prev = 0;
    for (int round = 0; round < DT_HUF_MAX_LEN; round++) {
        /* merge the leaves with the packages (pairs) of the previous round, by weight; leaves first on ties */
        const int n_pack = n_prev / 2;
        int li = 0, pi = 0, k = 0;
        while (li < n || pi < n_pack) {
            const uint64_t pw = pi < n_pack ? prev[2 * pi].w + prev[2 * pi + 1].w : 0;
            if (pi >= n_pack || (li < n && leaf[li][0] <= pw)) {
                memset(cur[k].cnt, 0, 256);
                cur[k].w = leaf[li][0];
                cur[k].cnt[leaf[li][1]] = 1;
                li++;
            } else {
                cur[k].w = pw;
                for (int s = 0; s < 256; s++) cur[k].cnt[s] = (uint8_t)(prev[2 * pi].cnt[s] + prev[2 * pi + 1].cnt[s]);
                pi++;
            }
            k++;
        }
        pm_item_t* t = prev;
        prev = cur;
        cur = t;
        n_prev = k;
    }
    /* the 2n - 2 lightest items of the last round */
    for (int i = 0; i < 2 * n - 2 && i < n_prev; i++)
        for (int s = 0; s < 256; s++) len_out[s] = (uint8_t)(len_out[s] + prev[i].cnt[s]);
    free(prev);
    free(cur);
    return ZXC_OK;
}

/* literals of one encoded block -> freq (the block formats: SURVEY.md appendix A; RLE tokens: src/lib/zxc_decompress.c:906-975) */
static void dt_count_block_literals(const uint8_t* blk, uint32_t size, const uint8_t* raw_src, size_t raw_len, uint32_t* freq) {
    if (size < BLK_HDR) return;
    const uint32_t csz = rd32(blk + 3);
    if ((uint64_t)BLK_HDR + csz > size) return;
    const uint8_t* p = blk + BLK_HDR;
    if (blk[0] == BLK_RAW) { /* stored: every byte of the slice is a literal */
        for (size_t i = 0; i < raw_len; i++) freq[raw_src[i]]++;
        return;
    }
    if (blk[0] != BLK_GLO || csz < 12) return;
    const uint32_t n_lit = rd32(p + 4);
    const uint8_t enc_lit = p[8], enc_tok = p[9];
    uint32_t at = 12, lit_bytes = n_lit;
    if (enc_lit != 0) { if (at + 4 > csz) return; lit_bytes = rd32(p + at); at += 4; }
    if (enc_tok == 2) at += 4;
    if ((uint64_t)at + lit_bytes > csz) return;
    const uint8_t* l = p + at;
    if (enc_lit == 0) {
        for (uint32_t i = 0; i < lit_bytes; i++) freq[l[i]]++;
    } else if (enc_lit == 1) {
        uint32_t i = 0;
        while (i < lit_bytes) {
            const uint8_t t = l[i++];
            if (t < 0x80) {
                const uint32_t k = (uint32_t)t + 1;
                for (uint32_t j = 0; j < k && i < lit_bytes; j++) freq[l[i++]]++;
            } else if (i < lit_bytes) {
                freq[l[i++]] += (uint32_t)(t & 0x7F) + 4;
            }
        }
    }
}

int zxc_train_dict_huf(const void* const* samples, const size_t* sample_sizes, const size_t n_samples, const void* dict,
                       const size_t dict_size, uint8_t* huf_lengths_out) {
    if (!samples || !sample_sizes || n_samples == 0 || !dict || dict_size == 0 || !huf_lengths_out) return ZXC_ERROR_NULL_INPUT;
    if (dict_size > ZXC_DICT_SIZE_MAX) return ZXC_ERROR_DICT_TOO_LARGE;
    /* the slices: every one while the corpus fits the budget, else one out of `every`, spread over all samples (:527-548) */
    size_t corpus = 0, n_slices = 0;
    for (size_t s = 0; s < n_samples; s++) corpus += sample_sizes[s];
    const size_t every = corpus > DT_HUF_BUDGET ? (corpus + DT_HUF_BUDGET - 1) / DT_HUF_BUDGET : 1;
    size_t idx = 0;
    for (size_t s = 0; s < n_samples; s++) {
        if (!samples[s] || sample_sizes[s] == 0) continue;
        for (size_t off = 0; off < sample_sizes[s]; off += DT_HUF_SLICE, idx++) n_slices += idx % every == 0;
    }
    uint32_t freq[256];
    memset(freq, 0, sizeof freq);
    int rc = ZXC_OK;
    if (n_slices > 0) {
        if (zxc_mi355x_device_count() <= 0) return ZXC_ERROR_GPU_UNAVAILABLE;
        /* one source image, a slice per 4 KiB block. The encoder takes blocks of one size; a shorter slice (a sample's tail, a small
         * sample) is continued with itself, cyclically: the parser covers that with one match at distance = the slice's length and
         * no literal is added to the histogram */
        const uint32_t bs = DT_HUF_SLICE;
        uint8_t* img = (uint8_t*)malloc(n_slices * (size_t)bs);
        if (!img) return ZXC_ERROR_MEMORY;
        size_t k = 0;
        idx = 0;
        for (size_t s = 0; s < n_samples; s++) {
            if (!samples[s] || sample_sizes[s] == 0) continue;
            const uint8_t* sp = (const uint8_t*)samples[s];
            for (size_t off = 0; off < sample_sizes[s]; off += DT_HUF_SLICE, idx++) {
                if (idx % every) continue;
                const size_t len = sample_sizes[s] - off < DT_HUF_SLICE ? sample_sizes[s] - off : DT_HUF_SLICE;
                memcpy(img + k * bs, sp + off, len);
                for (size_t j = len; j < bs; j++) img[k * bs + j] = img[k * bs + j - len];
                k++;
            }
        }
        const size_t total = n_slices * (size_t)bs;
        const uint32_t stride = zxc_mi355x_encode_slot_stride(bs);
        void* d_src = zxc_mi355x_malloc(total + 64);
        void* d_dict = zxc_mi355x_malloc(dict_size + 64);
        void* d_work = zxc_mi355x_malloc((size_t)zxc_mi355x_encode_dict_work_size(total, bs, (uint32_t)dict_size));
        void* d_slots = zxc_mi355x_malloc(n_slices * (size_t)stride);
        void* d_sizes = zxc_mi355x_malloc(n_slices * 4);
        uint8_t* slots = (uint8_t*)malloc(n_slices * (size_t)stride);
        uint32_t* sizes = (uint32_t*)malloc(n_slices * 4);
        rc = (d_src && d_dict && d_work && d_slots && d_sizes && slots && sizes) ? ZXC_OK : ZXC_ERROR_MEMORY;
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_h2d(d_src, img, total);
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_h2d(d_dict, dict, dict_size);
        if (rc == ZXC_OK)
            rc = zxc_mi355x_encode_blocks_dict_device(d_src, total, bs, 5, 0, d_dict, (uint32_t)dict_size, d_work, d_slots, (uint32_t*)d_sizes, NULL);
        if (rc == ZXC_OK) rc = zxc_mi355x_synchronize(NULL);
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(sizes, d_sizes, n_slices * 4);
        if (rc == ZXC_OK) rc = zxc_mi355x_memcpy_d2h(slots, d_slots, n_slices * (size_t)stride);
        for (size_t i = 0; rc == ZXC_OK && i < n_slices; i++) {
            if (sizes[i] > stride) { rc = ZXC_ERROR_CORRUPT_DATA; break; }
            dt_count_block_literals(slots + i * (size_t)stride, sizes[i], img + i * (size_t)bs, bs, freq);
        }
        zxc_mi355x_free(d_src); zxc_mi355x_free(d_dict); zxc_mi355x_free(d_work); zxc_mi355x_free(d_slots); zxc_mi355x_free(d_sizes);
        free(slots); free(sizes); free(img);
        if (rc != ZXC_OK) return rc;
    }
    uint32_t any = 0;
    for (int i = 0; i < 256; i++) any |= freq[i];
    if (!any) { /* no literals left (a low-entropy corpus): an all-zero table, every block keeps its own (:566-573) */
        memset(huf_lengths_out, 0, ZXC_HUF_TABLE_SIZE);
        return ZXC_OK;
    }
    uint8_t len[256];
    rc = huf_lengths_limited(freq, len);
    if (rc != ZXC_OK) return rc;
    for (int i = 0; i < 256; i += 2) huf_lengths_out[i >> 1] = (uint8_t)((len[i] & 0x0F) | ((len[i + 1] & 0x0F) << 4)); /* :952-958 */
    return ZXC_OK;
}

int64_t zxc_dict_train(const void* const* samples, const size_t* sample_sizes, const size_t n_samples, void* zxd_buf,
                       const size_t zxd_capacity) {
    if (!samples || !sample_sizes || n_samples == 0 || !zxd_buf || zxd_capacity == 0) return ZXC_ERROR_NULL_INPUT;
    uint8_t* content = (uint8_t*)malloc(ZXC_DICT_SIZE_MAX);
    if (!content) return ZXC_ERROR_MEMORY;
    int64_t out;
    const int64_t n = zxc_train_dict(samples, sample_sizes, n_samples, content, ZXC_DICT_SIZE_MAX);
    if (n <= 0) out = n < 0 ? n : (int64_t)ZXC_ERROR_SRC_TOO_SMALL;
    else {
        uint8_t huf[ZXC_HUF_TABLE_SIZE];
        const int hrc = zxc_train_dict_huf(samples, sample_sizes, n_samples, content, (size_t)n, huf);
        out = hrc != ZXC_OK ? (int64_t)hrc : zxc_dict_save(content, (size_t)n, huf, zxd_buf, zxd_capacity);
    }
    free(content);
    return out;
}
