/*
 * zxc_oracle.c — TEST INFRASTRUCTURE ONLY (see zxc_oracle.h).
 *
 * A from-scratch, exact-copy (no wild copies, no SIMD) restatement of the ZXC
 * v8 decode path. Each function cites the reference lines whose behaviour it
 * restates. Structure is deliberately different from the reference: one
 * sequence loop with exact bounds, a recursive top-down PivCo rebuild, and
 * byte-at-a-time match copies — this is the executable spec the HIP kernels are
 * checked against, not an optimised decoder.
 *
 * Semantics for malformed input follow SURVEY.md Appendix A.2: the reference's
 * SAFE/FAST loop phases only matter for which code a doubly-broken block
 * reports; this restatement reports errors in sequence order.
 */
#include "zxc_oracle.h"

#include <stdlib.h>
#include <string.h>

#define ZXO_MAGIC 0x9CB02EF5u
#define ZXO_VERSION 8
#define ZXO_FILE_HDR 16
#define ZXO_BLK_HDR 8
#define ZXO_FOOTER 12
#define ZXO_TAIL_PAD 2112 /* ZXC_DECOMPRESS_TAIL_PAD = 32*66, zxc_internal.h:341 */
#define ZXO_LIT_SLACK 32  /* ZXC_BLOCK_LIT_SLACK, zxc_internal.h:339 */
#define ZXO_MIN_MATCH 5

enum { BLK_RAW = 0, BLK_GLO = 1, BLK_GHI = 2, BLK_SEK = 254, BLK_EOF = 255 };

static uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t le32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

/* ------------------------------------------------------------------------- */
/* header hashes: zxc_hash8 / zxc_hash16 (src/lib/zxc_internal.h:1188-1214)  */
/* ------------------------------------------------------------------------- */
static uint64_t xorshift_mix(uint64_t h) {
    h ^= h << 13;
    h ^= h >> 7;
    h ^= h << 17;
    return h;
}
uint8_t zxo_hash8(const uint8_t p[8]) {
    const uint64_t h = xorshift_mix(le64(p) ^ 0x9E3779B97F4A7C15ull);
    return (uint8_t)((h >> 32) ^ h);
}
uint16_t zxo_hash16(const uint8_t p[16]) {
    const uint64_t h = xorshift_mix(le64(p) ^ le64(p + 8) ^ 0xD2D84A61D2D84A61ull);
    const uint32_t r = (uint32_t)((h >> 32) ^ h);
    return (uint16_t)((r >> 16) ^ r);
}

/* ------------------------------------------------------------------------- */
/* rapidhash v3 (vendored by the reference: src/lib/vendors/rapidhash.h;     */
/* block checksum = 64-bit hash folded hi^lo, src/lib/zxc_internal.h:1353).  */
/* Restated from the published algorithm: 7 independent 16-byte lanes per    */
/* 112-byte stripe, 128-bit multiply-fold ("mum") mixing.                    */
/* ------------------------------------------------------------------------- */
static const uint64_t RS[8] = {0x2d358dccaa6c78a5ull, 0x8bb84b93962eacc9ull, 0x4b33a62ed433d4a3ull,
                               0x4d5a2da51de1aa47ull, 0xa0761d6478bd642full, 0xe7037ed1a0b428dbull,
                               0x90ed1765281c388cull, 0xaaaaaaaaaaaaaaaaull};
static void mum(uint64_t* a, uint64_t* b) {
    const __uint128_t r = (__uint128_t)(*a) * (*b);
    *a = (uint64_t)r;
    *b = (uint64_t)(r >> 64);
}
static uint64_t mix(uint64_t a, uint64_t b) {
    mum(&a, &b);
    return a ^ b;
}
uint64_t zxo_rapidhash(const void* data, size_t len, uint64_t seed) {
    const uint8_t* p = (const uint8_t*)data;
    uint64_t a = 0, b = 0;
    size_t i = len;
    seed ^= mix(seed ^ RS[2], RS[1]);
    if (len <= 16) {
        if (len >= 4) {
            seed ^= len;
            if (len >= 8) {
                a = le64(p);
                b = le64(p + len - 8);
            } else {
                a = le32(p);
                b = le32(p + len - 4);
            }
        } else if (len > 0) {
            a = ((uint64_t)p[0] << 45) | p[len - 1];
            b = p[len >> 1];
        }
    } else {
        if (len > 112) {
            uint64_t s[7];
            for (int k = 0; k < 7; k++) s[k] = seed;
            do {
                for (int k = 0; k < 7; k++)
                    s[k] = mix(le64(p + 16 * k) ^ RS[k], le64(p + 16 * k + 8) ^ s[k]);
                p += 112;
                i -= 112;
            } while (i > 112);
            /* fold: seed^=s1; s2^=s3; s4^=s5; seed^=s6; s2^=s4; seed^=s2 */
            seed = s[0] ^ s[1] ^ s[6] ^ (s[2] ^ s[3] ^ s[4] ^ s[5]);
        }
        if (i > 16) {
            static const int sel[6] = {2, 2, 1, 1, 2, 1};
            for (int k = 0; k < 6 && i > (size_t)(16 * (k + 1)); k++)
                seed = mix(le64(p + 16 * k) ^ RS[sel[k]], le64(p + 16 * k + 8) ^ seed);
        }
        a = le64(p + i - 16) ^ i;
        b = le64(p + i - 8);
    }
    a ^= RS[1];
    b ^= seed;
    mum(&a, &b);
    return mix(a ^ RS[7], b ^ RS[1] ^ i);
}
uint32_t zxo_checksum32(const void* data, size_t len) {
    const uint64_t h = zxo_rapidhash(data, len, 0);
    return (uint32_t)(h ^ (h >> 32));
}
/* zxc_dict_id (src/lib/zxc_dict.c:35-45) */
static uint32_t dict_id_of(const uint8_t* dict, size_t n, const uint8_t* huf) {
    if (!dict || n == 0) return 0;
    const uint32_t base = zxo_checksum32(dict, n);
    if (!huf) return base;
    const uint64_t h = zxo_rapidhash(huf, 128, base);
    return (uint32_t)(h ^ (h >> 32));
}

/* ------------------------------------------------------------------------- */
/* container headers (src/lib/zxc_common.c:574-603, :638-654)                */
/* ------------------------------------------------------------------------- */
int zxo_read_file_header(const uint8_t* src, size_t n, uint32_t* block_size, int* has_checksum,
                         uint32_t* dict_id) {
    if (n < ZXO_FILE_HDR) return ZXO_E_SRC_TOO_SMALL;
    if (le32(src) != ZXO_MAGIC) return ZXO_E_BAD_MAGIC;
    if (src[4] != ZXO_VERSION) return ZXO_E_BAD_VERSION;
    uint8_t t[16];
    memcpy(t, src, 16);
    t[14] = t[15] = 0;
    if (le16(src + 14) != zxo_hash16(t) || (src[6] & 0x0F) != 0) return ZXO_E_BAD_HEADER;
    if (src[5] < 12 || src[5] > 21) return ZXO_E_BAD_BLOCK_SIZE;
    if (block_size) *block_size = 1u << src[5];
    if (has_checksum) *has_checksum = (src[6] & 0x80) ? 1 : 0;
    if (dict_id) *dict_id = (src[6] & 0x40) ? le32(src + 7) : 0;
    return ZXO_OK;
}

typedef struct {
    uint8_t type;
    uint32_t comp_size;
} blk_hdr_t;

static int read_block_header(const uint8_t* src, size_t n, blk_hdr_t* bh) {
    if (n < ZXO_BLK_HDR) return ZXO_E_SRC_TOO_SMALL;
    uint8_t t[8];
    memcpy(t, src, 8);
    t[7] = 0;
    if (src[7] != zxo_hash8(t)) return ZXO_E_BAD_HEADER;
    bh->type = src[0];
    bh->comp_size = le32(src + 3);
    return ZXO_OK;
}

/* ------------------------------------------------------------------------- */
/* prefix varint (src/lib/zxc_decompress.c:51-88): a bad or truncated varint */
/* yields 0 and parks the cursor at the end, so every later read yields 0.   */
/* ------------------------------------------------------------------------- */
static uint32_t read_varint(const uint8_t** pp, const uint8_t* end) {
    const uint8_t* p = *pp;
    if (p >= end) return 0;
    const uint32_t b0 = p[0];
    if (b0 < 0x80) {
        *pp = p + 1;
        return b0;
    }
    if (b0 < 0xC0) {
        if (p + 1 >= end) { *pp = end; return 0; }
        *pp = p + 2;
        return (b0 & 0x3F) | ((uint32_t)p[1] << 6);
    }
    if (b0 < 0xE0) {
        if (p + 2 >= end) { *pp = end; return 0; }
        *pp = p + 3;
        return (b0 & 0x1F) | ((uint32_t)p[1] << 5) | ((uint32_t)p[2] << 13);
    }
    *pp = end;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* PivCo Huffman section (docs/FORMAT.md §5.2.1; src/lib/zxc_huffman.c       */
/* :960-977 lengths, :1042-1170 tree + flat roots, :2271-2430 decode core).  */
/* Top-down recursive rebuild instead of the reference's bottom-up ping-pong.*/
/* ------------------------------------------------------------------------- */
#define HUF_MAXLEN 11
#define HUF_NODES 511

typedef struct {
    int16_t child[2];
    int16_t sym; /* >= 0 for a leaf */
    uint8_t flat_d, covered;
    uint32_t count;
    const uint8_t* run;
} hnode_t;

typedef struct {
    hnode_t nd[HUF_NODES];
    int n_nodes;
    int16_t bfs[HUF_NODES];
} htree_t;

static int huf_tree_build(const uint8_t code_len[256], htree_t* t) {
    uint32_t cnt[HUF_MAXLEN + 2] = {0};
    int present = 0;
    for (int s = 0; s < 256; s++) {
        if (!code_len[s]) continue;
        if (code_len[s] > HUF_MAXLEN) return -1;
        cnt[code_len[s]]++;
        present++;
    }
    if (!present) return -1;
    if (present >= 2) {
        uint32_t kraft = 0;
        for (int l = 1; l <= HUF_MAXLEN; l++) kraft += cnt[l] << (HUF_MAXLEN - l);
        if (kraft != (1u << HUF_MAXLEN)) return -1;
    } else if (cnt[1] != 1) {
        return -1;
    }
    uint32_t next[HUF_MAXLEN + 2] = {0};
    uint32_t code = 0;
    for (int l = 1; l <= HUF_MAXLEN; l++) {
        code = (code + cnt[l - 1]) << 1;
        next[l] = code;
    }
    t->n_nodes = 1;
    t->nd[0].child[0] = t->nd[0].child[1] = -1;
    t->nd[0].sym = -1;
    for (int s = 0; s < 256; s++) {
        const int l = code_len[s];
        if (!l) continue;
        const uint32_t c = next[l]++;
        if (c >> l) return -1;
        int cur = 0;
        for (int d = l - 1; d >= 0; d--) {
            if (t->nd[cur].sym >= 0) return -1;
            const int bit = (c >> d) & 1;
            int nx = t->nd[cur].child[bit];
            if (nx < 0) {
                if (t->n_nodes >= HUF_NODES) return -1;
                nx = t->n_nodes++;
                t->nd[nx].child[0] = t->nd[nx].child[1] = -1;
                t->nd[nx].sym = -1;
                t->nd[cur].child[bit] = (int16_t)nx;
            }
            cur = nx;
        }
        if (t->nd[cur].child[0] >= 0 || t->nd[cur].child[1] >= 0) return -1;
        t->nd[cur].sym = (int16_t)s;
    }
    /* BFS order = wire order */
    int head = 0, tail = 0;
    t->bfs[tail++] = 0;
    while (head < tail) {
        const hnode_t* nd = &t->nd[t->bfs[head++]];
        for (int b = 0; b < 2; b++)
            if (nd->child[b] >= 0) t->bfs[tail++] = nd->child[b];
    }
    /* flat roots: maximal complete subtrees of depth >= 2 */
    int8_t mn[HUF_NODES], mx[HUF_NODES];
    for (int i = t->n_nodes - 1; i >= 0; i--) {
        const int id = t->bfs[i];
        const hnode_t* nd = &t->nd[id];
        if (nd->sym >= 0) {
            mn[id] = mx[id] = 0;
        } else if (nd->child[0] >= 0 && nd->child[1] >= 0) {
            const int a = mn[nd->child[0]], b = mn[nd->child[1]];
            const int c = mx[nd->child[0]], d = mx[nd->child[1]];
            mn[id] = (int8_t)(1 + (a < b ? a : b));
            mx[id] = (int8_t)(1 + (c > d ? c : d));
        } else {
            mn[id] = 0;
            mx[id] = HUF_MAXLEN;
        }
    }
    t->nd[0].covered = 0;
    for (int i = 0; i < t->n_nodes; i++) {
        const int id = t->bfs[i];
        hnode_t* nd = &t->nd[id];
        nd->flat_d = 0;
        if (!nd->covered && nd->sym < 0 && mn[id] == mx[id] && mn[id] >= 2) nd->flat_d = (uint8_t)mn[id];
        const uint8_t cov = (uint8_t)(nd->covered || nd->flat_d);
        for (int b = 0; b < 2; b++)
            if (nd->child[b] >= 0) t->nd[nd->child[b]].covered = cov;
    }
    return 0;
}

static uint32_t popcount_bits(const uint8_t* p, uint32_t nbits) {
    uint32_t ones = 0;
    for (uint32_t k = 0; k < nbits; k++) ones += (p[k >> 3] >> (k & 7)) & 1;
    return ones;
}

/* leaf symbol reached from `id` by following `code` LSB-first for `d` levels */
static uint8_t flat_symbol(const htree_t* t, int id, uint32_t code, int d) {
    for (int l = 0; l < d; l++) id = t->nd[id].child[(code >> l) & 1];
    return (uint8_t)t->nd[id].sym;
}

static int huf_rebuild(const htree_t* t, int id, uint8_t* out) {
    const hnode_t* nd = &t->nd[id];
    const uint32_t c = nd->count;
    if (c == 0) return 0;
    if (nd->sym >= 0) {
        memset(out, nd->sym, c);
        return 0;
    }
    if (nd->flat_d) {
        const int D = nd->flat_d;
        for (uint32_t i = 0; i < c; i++) {
            uint32_t code = 0;
            const uint64_t bit0 = (uint64_t)i * D;
            for (int l = 0; l < D; l++) {
                const uint64_t bp = bit0 + l;
                code |= (uint32_t)((nd->run[bp >> 3] >> (bp & 7)) & 1) << l;
            }
            out[i] = flat_symbol(t, id, code, D);
        }
        return 0;
    }
    const int c0 = nd->child[0], c1 = nd->child[1];
    const uint32_t nr = (c1 >= 0) ? t->nd[c1].count : 0;
    const uint32_t nl = c - nr;
    uint8_t* L = (uint8_t*)malloc((size_t)nl + 1);
    uint8_t* R = (uint8_t*)malloc((size_t)nr + 1);
    if (!L || !R) { free(L); free(R); return ZXO_E_MEMORY; }
    int rc = 0;
    if (c0 >= 0) rc = huf_rebuild(t, c0, L);
    if (!rc && c1 >= 0) rc = huf_rebuild(t, c1, R);
    if (!rc) {
        uint32_t lp = 0, rp = 0;
        for (uint32_t k = 0; k < c; k++)
            out[k] = ((nd->run[k >> 3] >> (k & 7)) & 1) ? R[rp++] : L[lp++];
    }
    free(L);
    free(R);
    return rc;
}

static int huf_decode_core(const uint8_t* payload, size_t psize, uint8_t* dst, size_t n, htree_t* t) {
    if (n == 0) return ZXO_E_CORRUPT_DATA;
    const uint8_t* p = payload;
    const uint8_t* const pend = payload + psize;
    for (int i = 0; i < t->n_nodes; i++) t->nd[i].count = 0;
    t->nd[0].count = (uint32_t)n;
    for (int i = 0; i < t->n_nodes; i++) {
        hnode_t* nd = &t->nd[t->bfs[i]];
        if (nd->covered || nd->sym >= 0) continue;
        const uint32_t c = nd->count;
        const size_t nbytes = nd->flat_d ? ((size_t)c * nd->flat_d + 7) / 8 : ((size_t)c + 7) / 8;
        if ((size_t)(pend - p) < nbytes) return ZXO_E_CORRUPT_DATA;
        nd->run = p;
        p += nbytes;
        if (nd->flat_d) continue;
        const uint32_t ones = popcount_bits(nd->run, c);
        if (nd->child[1] >= 0) t->nd[nd->child[1]].count = ones;
        else if (ones) return ZXO_E_CORRUPT_DATA;
        if (nd->child[0] >= 0) t->nd[nd->child[0]].count = c - ones;
        else if (c - ones) return ZXO_E_CORRUPT_DATA;
    }
    return huf_rebuild(t, 0, dst);
}

static int huf_unpack_lengths(const uint8_t* in, uint8_t code_len[256]) {
    int present = 0, maxl = 0;
    for (int i = 0; i < 256; i++) {
        const uint8_t v = (i & 1) ? (in[i >> 1] >> 4) : (in[i >> 1] & 0x0F);
        code_len[i] = v;
        if (v > maxl) maxl = v;
        if (v) present++;
    }
    return (maxl > HUF_MAXLEN || !present) ? ZXO_E_CORRUPT_DATA : ZXO_OK;
}

int zxo_huf_decode_section(const uint8_t* payload, size_t psize, uint8_t* dst, size_t n) {
    if (psize < 128) return ZXO_E_CORRUPT_DATA;
    uint8_t code_len[256];
    if (huf_unpack_lengths(payload, code_len) != ZXO_OK) return ZXO_E_CORRUPT_DATA;
    htree_t* t = (htree_t*)malloc(sizeof(htree_t));
    if (!t) return ZXO_E_MEMORY;
    int rc = ZXO_E_CORRUPT_DATA;
    if (huf_tree_build(code_len, t) == 0) rc = huf_decode_core(payload + 128, psize - 128, dst, n, t);
    free(t);
    return rc;
}

/* shared-table section (enc_lit = 3): no inline lengths (zxc_huffman.c:2467) */
static int huf_decode_section_dict(const uint8_t* payload, size_t psize, uint8_t* dst, size_t n,
                                   const uint8_t* dict_huf) {
    uint8_t code_len[256];
    if (!dict_huf) return ZXO_E_DICT_REQUIRED;
    if (huf_unpack_lengths(dict_huf, code_len) != ZXO_OK) return ZXO_E_DICT_REQUIRED;
    htree_t* t = (htree_t*)malloc(sizeof(htree_t));
    if (!t) return ZXO_E_MEMORY;
    int rc = ZXO_E_DICT_REQUIRED;
    if (huf_tree_build(code_len, t) == 0) rc = huf_decode_core(payload, psize, dst, n, t);
    free(t);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* RLE literal section (src/lib/zxc_decompress.c:906-975)                    */
/* ------------------------------------------------------------------------- */
static int rle_expand(const uint8_t* r, size_t rsize, uint8_t* w, size_t n) {
    const uint8_t* const r_end = r + rsize;
    uint8_t* const w_end = w + n;
    while (r < r_end && w < w_end) {
        const uint8_t tok = *r++;
        if (!(tok & 0x80)) {
            const uint32_t len = (uint32_t)tok + 1;
            if ((size_t)(w_end - w) < len || (size_t)(r_end - r) < len) return ZXO_E_CORRUPT_DATA;
            memcpy(w, r, len);
            w += len;
            r += len;
        } else {
            const uint32_t len = (tok & 0x7F) + 4;
            if ((size_t)(w_end - w) < len || r >= r_end) return ZXO_E_CORRUPT_DATA;
            memset(w, *r++, len);
            w += len;
        }
    }
    return (w == w_end) ? ZXO_OK : ZXO_E_CORRUPT_DATA;
}

/* ------------------------------------------------------------------------- */
/* sequence executor shared by GLO and GHI. `win` points at the first output */
/* byte; win[-dict_size .. -1] holds the dictionary prefix.                  */
/* (GLO: src/lib/zxc_decompress.c:1025-1208; GHI: :1271-1468; A.2/A.3)       */
/* ------------------------------------------------------------------------- */
typedef struct {
    const uint8_t* lit;
    size_t n_lit;
    const uint8_t* extras;
    const uint8_t* extras_end;
    uint8_t* win;
    size_t cap;
    size_t dict_size;
    size_t p, lp;
    int fourx; /* the reference is still in its 4x-unrolled loops (see reserve_4x) */
} seq_state_t;

/* The reference's accept / reject decisions are those of an exact decoder EXCEPT inside its 4x-unrolled loops
 * (GLO src/lib/zxc_decompress.c:1050-1103, GHI :1296-1384; non-"safe" variants only): there a sequence with a
 * varint-extended length is refused with OVERFLOW unless the worst-case inline output of the batch's remaining
 * sequences + ZXC_PAD_SIZE still fits behind it, and its literals + the remaining sequences' raw literal fields fit
 * the literal stream (DECODE_GLO_SEQ :626-656, DECODE_GHI_SEQ :701-727). Which sequences run in 4x batches: a batch of
 * four starts at every multiple of 4 while n_seq >= 4 remain, d_ptr < d_end - D and l_ptr < l_end - Lm (Lm < sz_lit);
 * all three are monotone in the cursors, so the 4x batches are a PREFIX of the sequence list (the SAFE -> FAST switch
 * at d_bounds keeps the grouping; GHI's 1x offset-validating loop :1322-1357 only runs once the 4x condition has failed
 * for good). `W` = worst inline output per sequence (33 GLO / 513 GHI), D = 168 / 2112, Lm = 56 / 1016.
 * Called in front of every sequence i with its final ll / ml (ml includes the minimum match length);
 * `raw_ll_rest` = sum of the RAW literal-length fields of the sequences behind i in its batch. */
static int reserve_4x(seq_state_t* s, uint32_t i, uint32_t n_seq, int esc_l, int esc_m, uint64_t ll, uint64_t ml,
                      uint64_t raw_ll_rest, uint32_t W, uint32_t D, uint32_t Lm) {
    if (!s->fourx) return ZXO_OK;
    if ((i & 3u) == 0u)
        s->fourx = (n_seq - i >= 4u) && ((uint64_t)s->p + D < s->cap) && (s->n_lit > Lm) && (s->lp < s->n_lit - Lm);
    if (!s->fourx || !(esc_l || esc_m)) return ZXO_OK;
    const uint32_t n_rem = 3u - (i & 3u);
    if (esc_l && ll + raw_ll_rest > s->n_lit - s->lp) return ZXO_E_OVERFLOW;
    /* (:645-648 with the raw ml, :657-660 with the extended one: the union is the test on the final lengths) */
    if (ll + ml + (uint64_t)n_rem * W + 32u > s->cap - s->p) return ZXO_E_OVERFLOW;
    return ZXO_OK;
}

static int emit_sequence(seq_state_t* s, uint64_t ll, uint64_t ml, uint32_t off) {
    if (ll + ml > s->cap - s->p || ll > s->n_lit - s->lp) return ZXO_E_OVERFLOW;
    memcpy(s->win + s->p, s->lit + s->lp, (size_t)ll);
    s->lp += (size_t)ll;
    s->p += (size_t)ll;
    if ((uint64_t)s->p + s->dict_size < off) return ZXO_E_BAD_OFFSET;
    uint8_t* d = s->win + s->p;
    for (uint64_t k = 0; k < ml; k++) d[k] = d[(ptrdiff_t)k - (ptrdiff_t)off];
    s->p += (size_t)ml;
    return ZXO_OK;
}

static int finish_block(seq_state_t* s) {
    const size_t rem = s->n_lit - s->lp;
    if (rem > s->cap - s->p) return ZXO_E_OVERFLOW;
    memcpy(s->win + s->p, s->lit + s->lp, rem);
    s->p += rem;
    return (int)s->p;
}

static int decode_glo(const zxo_ctx_t* ctx, const uint8_t* src, size_t n, uint8_t* win, size_t cap,
                      zxo_block_stats_t* st) {
    if (n < 12) return ZXO_E_BAD_HEADER;
    const uint32_t n_seq = le32(src), n_lit = le32(src + 4);
    const uint8_t enc_lit = src[8], enc_tok = src[9], enc_off = src[11];
    const size_t desc = (enc_lit != 0 ? 4 : 0) + (enc_tok == 2 ? 4 : 0);
    if (n < 12 + desc) return ZXO_E_BAD_HEADER;
    const uint8_t* d = src + 12;
    uint32_t lit_comp = n_lit, tok_comp = n_seq;
    if (enc_lit != 0) { lit_comp = le32(d); d += 4; }
    if (enc_tok == 2) { tok_comp = le32(d); d += 4; }
    if (enc_off > 1) return ZXO_E_CORRUPT_DATA;

    const uint8_t* const pdata = src + 12 + desc;
    const size_t avail = n - 12 - desc;
    uint8_t* lit_buf = NULL; /* decoded literals when enc_lit != 0 */
    uint8_t* tok_buf = NULL;
    const uint8_t* lit = pdata;
    size_t lit_n = n_lit;
    int rc = ZXO_OK;

    if (enc_lit == 2 || enc_lit == 3) {
        if (lit_comp > avail) return ZXO_E_CORRUPT_DATA;
        if (n_lit == 0) {
            lit_n = 0;
        } else {
            if (n_lit > cap) return ZXO_E_DST_TOO_SMALL;
            if (enc_lit == 3 && !ctx->dict_huf) return ZXO_E_DICT_REQUIRED;
            if (n_lit > ctx->block_size) return ZXO_E_CORRUPT_DATA;
            lit_buf = (uint8_t*)malloc(n_lit);
            if (!lit_buf) return ZXO_E_MEMORY;
            rc = (enc_lit == 2) ? zxo_huf_decode_section(pdata, lit_comp, lit_buf, n_lit)
                                : huf_decode_section_dict(pdata, lit_comp, lit_buf, n_lit, ctx->dict_huf);
            lit = lit_buf;
        }
    } else if (enc_lit == 1) {
        if (n_lit > 0) {
            if (n_lit > cap) return ZXO_E_DST_TOO_SMALL;
            if (n_lit > ctx->block_size || lit_comp > avail) return ZXO_E_CORRUPT_DATA;
            lit_buf = (uint8_t*)malloc(n_lit);
            if (!lit_buf) return ZXO_E_MEMORY;
            rc = rle_expand(pdata, lit_comp, lit_buf, n_lit);
            lit = lit_buf;
        } else {
            lit_n = 0;
        }
    } else if (enc_lit != 0) {
        return ZXO_E_CORRUPT_DATA;
    }
    if (rc != ZXO_OK) goto out;

    {
        const uint64_t sz_off = (uint64_t)n_seq * (enc_off ? 1 : 2);
        const uint64_t consumed = (uint64_t)lit_comp + tok_comp + sz_off;
        if (consumed > avail || avail - lit_comp < ZXO_LIT_SLACK) { rc = ZXO_E_CORRUPT_DATA; goto out; }
        const uint8_t* tok = pdata + lit_comp;
        const uint8_t* offs = tok + tok_comp;
        const uint8_t* ext = offs + (size_t)sz_off;
        const uint8_t* const ext_end = src + n;
        if (enc_tok == 2) {
            if (n_seq > (size_t)ctx->block_size / 5 + 16 || n_seq > ctx->block_size) { rc = ZXO_E_CORRUPT_DATA; goto out; }
            tok_buf = (uint8_t*)malloc((size_t)n_seq + 1);
            if (!tok_buf) { rc = ZXO_E_MEMORY; goto out; }
            rc = zxo_huf_decode_section(tok, tok_comp, tok_buf, n_seq);
            if (rc != ZXO_OK) goto out;
            tok = tok_buf;
        } else if (enc_tok != 0) {
            rc = ZXO_E_CORRUPT_DATA;
            goto out;
        }
        if (st) {
            st->lit_bytes = lit_comp; st->tok_bytes = tok_comp; st->off_bytes = (uint32_t)sz_off;
            st->extra_bytes = (uint32_t)(ext_end - ext);
        }
        seq_state_t s = {lit, lit_n, ext, ext_end, win, cap, ctx->dict_size, 0, 0, !ctx->strict_tail};
        for (uint32_t i = 0; i < n_seq; i++) {
            uint64_t ll = tok[i] >> 4, ml = tok[i] & 15;
            const int esc_l = ll == 15, esc_m = ml == 15;
            const uint32_t off = 1u + (enc_off ? offs[i] : le16(offs + 2 * (size_t)i));
            if (ll == 15) { ll += read_varint(&s.extras, s.extras_end); if (st) st->n_varints++; }
            if (ml == 15) { ml += read_varint(&s.extras, s.extras_end); if (st) st->n_varints++; }
            ml += ZXO_MIN_MATCH;
            {
                uint64_t rest = 0;
                for (uint32_t j = i + 1; j <= (i | 3u) && j < n_seq; j++) rest += tok[j] >> 4;
                rc = reserve_4x(&s, i, n_seq, esc_l, esc_m, ll, ml, rest, 33u, 168u, 56u);
                if (rc != ZXO_OK) goto out;
            }
            if (st) {
                int lg = 0;
                while ((1u << lg) < off) lg++;
                st->off_hist[lg] += ml;
                st->match_bytes += ml;
                if (off < ml) st->overlap_matches++;
            }
            rc = emit_sequence(&s, ll, ml, off);
            if (rc != ZXO_OK) goto out;
        }
        rc = finish_block(&s);
    }
out:
    free(lit_buf);
    free(tok_buf);
    return rc;
}

static int decode_ghi(const zxo_ctx_t* ctx, const uint8_t* src, size_t n, uint8_t* win, size_t cap,
                      zxo_block_stats_t* st) {
    if (n < 12) return ZXO_E_BAD_HEADER;
    const uint32_t n_seq = le32(src), n_lit = le32(src + 4);
    if (src[8] != 0 || src[9] != 0) return ZXO_E_CORRUPT_DATA;
    const size_t avail = n - 12;
    const uint64_t consumed = (uint64_t)n_lit + (uint64_t)n_seq * 4;
    if (consumed > avail || avail - n_lit < ZXO_LIT_SLACK) return ZXO_E_CORRUPT_DATA;
    const uint8_t* lit = src + 12;
    const uint8_t* seqs = lit + n_lit;
    const uint8_t* ext = seqs + (size_t)n_seq * 4;
    if (st) { st->lit_bytes = n_lit; st->tok_bytes = n_seq * 4; st->extra_bytes = (uint32_t)(src + n - ext); }
    seq_state_t s = {lit, n_lit, ext, src + n, win, cap, ctx->dict_size, 0, 0, !ctx->strict_tail};
    for (uint32_t i = 0; i < n_seq; i++) {
        const uint32_t w = le32(seqs + 4 * (size_t)i);
        uint64_t ll = w >> 24;
        const uint32_t mb = (w >> 16) & 0xFF;
        uint64_t ml = mb + ZXO_MIN_MATCH;
        const int esc_l = ll == 255, esc_m = mb == 255;
        if (ll == 255) { ll += read_varint(&s.extras, s.extras_end); if (st) st->n_varints++; }
        if (mb == 255) { ml += read_varint(&s.extras, s.extras_end); if (st) st->n_varints++; }
        const uint32_t off = (w & 0xFFFF) + 1;
        {
            uint64_t rest = 0;
            for (uint32_t j = i + 1; j <= (i | 3u) && j < n_seq; j++) rest += seqs[4 * (size_t)j + 3];
            const int r4 = reserve_4x(&s, i, n_seq, esc_l, esc_m, ll, ml, rest, 513u, 2112u, 1016u);
            if (r4 != ZXO_OK) return r4;
        }
        if (st) {
            int lg = 0;
            while ((1u << lg) < off) lg++;
            st->off_hist[lg] += ml;
            st->match_bytes += ml;
            if (off < ml) st->overlap_matches++;
        }
        const int rc = emit_sequence(&s, ll, ml, off);
        if (rc != ZXO_OK) return rc;
    }
    return finish_block(&s);
}

/* zxc_decompress_chunk_wrapper_body (src/lib/zxc_decompress.c:1646-1695).
 * `win` must have dict_size readable bytes in front of it. */
static int decode_block_into(const zxo_ctx_t* ctx, const uint8_t* src, size_t src_sz, uint8_t* win,
                             size_t cap, zxo_block_stats_t* st) {
    if (src_sz < ZXO_BLK_HDR) return ZXO_E_SRC_TOO_SMALL;
    const uint8_t type = src[0];
    const uint32_t comp_sz = le32(src + 3);
    const size_t expect = (size_t)ZXO_BLK_HDR + comp_sz + (ctx->checksum_enabled ? 4 : 0);
    if (src_sz < expect) return ZXO_E_SRC_TOO_SMALL;
    const uint8_t* data = src + ZXO_BLK_HDR;
    if (ctx->checksum_enabled && le32(data + comp_sz) != zxo_checksum32(data, comp_sz))
        return ZXO_E_BAD_CHECKSUM;
    if (st) st->type = type;
    switch (type) {
        case BLK_GLO:
            if (st && comp_sz >= 12) {
                st->n_sequences = le32(data); st->n_literals = le32(data + 4);
                st->enc_lit = data[8]; st->enc_tok = data[9]; st->enc_off = data[11];
            }
            return decode_glo(ctx, data, comp_sz, win, cap, st);
        case BLK_GHI:
            if (st && comp_sz >= 12) { st->n_sequences = le32(data); st->n_literals = le32(data + 4); }
            return decode_ghi(ctx, data, comp_sz, win, cap, st);
        case BLK_RAW:
            if (comp_sz > cap) return ZXO_E_DST_TOO_SMALL;
            memcpy(win, data, comp_sz);
            return (int)comp_sz;
        case BLK_EOF:
            return ZXO_E_CORRUPT_DATA;
        default:
            return ZXO_E_BAD_BLOCK_TYPE;
    }
}

int zxo_decode_block(const zxo_ctx_t* ctx, const uint8_t* src, size_t src_sz, uint8_t* dst,
                     size_t dst_cap) {
    /* bounce through [dict | out] so offsets may reach into the dictionary */
    uint8_t* buf = (uint8_t*)malloc(ctx->dict_size + dst_cap + 1);
    if (!buf) return ZXO_E_MEMORY;
    if (ctx->dict_size) memcpy(buf, ctx->dict, ctx->dict_size);
    const int rc = decode_block_into(ctx, src, src_sz, buf + ctx->dict_size, dst_cap, NULL);
    if (rc > 0) memcpy(dst, buf + ctx->dict_size, (size_t)rc);
    free(buf);
    return rc;
}

int zxo_block_stats(const uint8_t* src, size_t src_sz, zxo_block_stats_t* st) {
    memset(st, 0, sizeof(*st));
    zxo_ctx_t ctx = {1u << 21, 0, NULL, 0, NULL};
    const size_t cap = (1u << 21) + ZXO_TAIL_PAD;
    uint8_t* buf = (uint8_t*)malloc(cap);
    if (!buf) return ZXO_E_MEMORY;
    const int rc = decode_block_into(&ctx, src, src_sz, buf, cap, st);
    free(buf);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* frame decode: zxc_decompress / zxc_decompress_frame                       */
/* (src/lib/zxc_dispatch.c:842-1005)                                         */
/* ------------------------------------------------------------------------- */
int64_t zxo_decompress(const void* src_v, size_t src_size, void* dst_v, size_t dst_capacity,
                       int checksum_enabled, const void* dict, size_t dict_size, const void* dict_huf) {
    const uint8_t* src = (const uint8_t*)src_v;
    uint8_t* dst = (uint8_t*)dst_v;
    if (!src || (!dst && dst_capacity != 0)) return ZXO_E_NULL_INPUT;
    if (src_size < ZXO_FILE_HDR + ZXO_FOOTER) return ZXO_E_SRC_TOO_SMALL;
    if (!dst || dst_capacity == 0) {
        if (le32(src) != ZXO_MAGIC) return ZXO_E_BAD_MAGIC;
        return le64(src + src_size - ZXO_FOOTER) == 0 ? 0 : ZXO_E_DST_TOO_SMALL;
    }
    uint32_t block_size = 0, hdr_dict_id = 0;
    int has_ck = 0;
    const int hrc = zxo_read_file_header(src, src_size, &block_size, &has_ck, &hdr_dict_id);
    if (hrc != ZXO_OK) return hrc;
    if (!dict) { dict_size = 0; dict_huf = NULL; }
    if (hdr_dict_id != 0) {
        if (!dict || dict_size == 0) return ZXO_E_DICT_REQUIRED;
        if (dict_id_of((const uint8_t*)dict, dict_size, (const uint8_t*)dict_huf) != hdr_dict_id)
            return ZXO_E_DICT_MISMATCH;
    }
    zxo_ctx_t ctx = {block_size, has_ck && checksum_enabled, (const uint8_t*)dict, dict_size,
                     (const uint8_t*)dict_huf};
    const size_t work = (size_t)block_size + ZXO_TAIL_PAD;
    uint8_t* bounce = (uint8_t*)malloc(dict_size + work);
    if (!bounce) return ZXO_E_MEMORY;
    if (dict_size) memcpy(bounce, dict, dict_size);
    uint8_t* const bwin = bounce + dict_size;

    const uint8_t* ip = src + ZXO_FILE_HDR;
    const uint8_t* const ip_end = src + src_size;
    size_t op = 0;
    uint32_t global = 0;
    int64_t ret = 0;
    for (;;) {
        if (ip >= ip_end) { ret = (int64_t)op; break; }
        const size_t rem = (size_t)(ip_end - ip);
        blk_hdr_t bh;
        if (read_block_header(ip, rem, &bh) != ZXO_OK) { ret = ZXO_E_BAD_HEADER; break; }
        if (bh.type == BLK_EOF) {
            if (bh.comp_size != 0) { ret = ZXO_E_BAD_HEADER; break; }
            const uint8_t* footer = src + src_size - ZXO_FOOTER;
            if (le64(footer) != (uint64_t)op) { ret = ZXO_E_CORRUPT_DATA; break; }
            if (ctx.checksum_enabled && le32(footer + 8) != global) { ret = ZXO_E_BAD_CHECKSUM; break; }
            ret = (int64_t)op;
            break;
        }
        const int res = decode_block_into(&ctx, ip, rem, bwin, work, NULL);
        if (res < 0) { ret = res; break; }
        if ((size_t)res > dst_capacity - op) { ret = ZXO_E_DST_TOO_SMALL; break; }
        memcpy(dst + op, bwin, (size_t)res);
        if (ctx.checksum_enabled)
            global = ((global << 1) | (global >> 31)) ^ le32(ip + ZXO_BLK_HDR + bh.comp_size);
        ip += (size_t)ZXO_BLK_HDR + bh.comp_size + (has_ck ? 4 : 0);
        op += (size_t)res;
    }
    free(bounce);
    return ret;
}

/* ------------------------------------------------------------------------- */
/* seek table (src/lib/zxc_seekable.c:270-396) and range decode (:695-785)   */
/* ------------------------------------------------------------------------- */
void zxo_seek_table_free(zxo_seek_table_t* t) {
    if (!t) return;
    free(t->comp_sizes);
    free(t->comp_offsets);
    t->comp_sizes = NULL;
    t->comp_offsets = NULL;
}

int zxo_seek_table_parse(const uint8_t* data, size_t n, zxo_seek_table_t* out) {
    memset(out, 0, sizeof(*out));
    if (!data || n < ZXO_FILE_HDR + 2 * ZXO_BLK_HDR + ZXO_FOOTER) return -1;
    if (zxo_read_file_header(data, n, &out->block_size, &out->has_checksum, &out->dict_id) != ZXO_OK)
        return -1;
    const uint64_t total = le64(data + n - ZXO_FOOTER);
    if (total == 0) return -1;
    const uint64_t nb = (total + out->block_size - 1) / out->block_size;
    if (nb > 0xFFFFFFFFull) return -1;
    const uint64_t entries = nb * 4;
    if (entries + ZXO_BLK_HDR + ZXO_FOOTER > n) return -1;
    const uint8_t* sek = data + n - ZXO_FOOTER - ZXO_BLK_HDR - (size_t)entries;
    blk_hdr_t bh;
    if (read_block_header(sek, ZXO_BLK_HDR + (size_t)entries, &bh) != ZXO_OK) return -1;
    if (bh.type != BLK_SEK || bh.comp_size != (uint32_t)entries) return -1;
    out->n_blocks = (uint32_t)nb;
    out->total_decomp = total;
    out->comp_sizes = (uint32_t*)calloc((size_t)nb, 4);
    out->comp_offsets = (uint64_t*)calloc((size_t)nb + 1, 8);
    if (!out->comp_sizes || !out->comp_offsets) { zxo_seek_table_free(out); return -1; }
    uint64_t acc = ZXO_FILE_HDR;
    for (uint32_t i = 0; i < out->n_blocks; i++) {
        const uint32_t cs = le32(sek + ZXO_BLK_HDR + 4 * (size_t)i);
        if (cs < ZXO_BLK_HDR || cs > n) { zxo_seek_table_free(out); return -1; }
        out->comp_sizes[i] = cs;
        out->comp_offsets[i] = acc;
        acc += cs;
        if (acc > n) { zxo_seek_table_free(out); return -1; }
    }
    out->comp_offsets[out->n_blocks] = acc;
    if (acc != (uint64_t)(sek - data) - ZXO_BLK_HDR) { zxo_seek_table_free(out); return -1; }
    if (read_block_header(data + acc, ZXO_BLK_HDR, &bh) != ZXO_OK || bh.type != BLK_EOF) {
        zxo_seek_table_free(out);
        return -1;
    }
    return 0;
}

int64_t zxo_seekable_decompress_range(const uint8_t* data, size_t n, const zxo_seek_table_t* t,
                                      void* dst_v, size_t dst_capacity, uint64_t offset, size_t len) {
    if (len == 0) return 0;
    if (!t || !dst_v) return ZXO_E_NULL_INPUT;
    if (dst_capacity < len) return ZXO_E_DST_TOO_SMALL;
    if (offset + len > t->total_decomp) return ZXO_E_SRC_TOO_SMALL;
    if (t->dict_id != 0) return ZXO_E_DICT_REQUIRED;
    /* seekable decode never verifies per-block checksums (zxc_seekable.c:707) */
    zxo_ctx_t ctx = {t->block_size, 0, NULL, 0, NULL};
    const size_t work = (size_t)t->block_size + ZXO_TAIL_PAD;
    uint8_t* buf = (uint8_t*)malloc(work);
    if (!buf) return ZXO_E_MEMORY;
    uint8_t* out = (uint8_t*)dst_v;
    size_t remaining = len;
    const uint32_t b0 = (uint32_t)(offset / t->block_size);
    const uint32_t b1 = (uint32_t)((offset + len - 1) / t->block_size);
    int64_t ret = (int64_t)len;
    for (uint32_t bi = b0; bi <= b1; bi++) {
        if (t->comp_offsets[bi] + t->comp_sizes[bi] > n) { ret = ZXO_E_SRC_TOO_SMALL; break; }
        const int res = decode_block_into(&ctx, data + t->comp_offsets[bi], t->comp_sizes[bi], buf, work, NULL);
        if (res < 0) { ret = res; break; }
        const uint64_t bstart = (uint64_t)bi * t->block_size;
        const size_t skip = offset > bstart ? (size_t)(offset - bstart) : 0;
        if ((size_t)res < skip) { ret = ZXO_E_CORRUPT_DATA; break; }
        const size_t av = (size_t)res - skip;
        const size_t cp = av < remaining ? av : remaining;
        memcpy(out, buf + skip, cp);
        out += cp;
        remaining -= cp;
    }
    free(buf);
    return ret;
}
