// zxc_decode_kernel.hip — ZXC v8 block decode for gfx950 (MI355X, CDNA4).
//
// One 64-lane wavefront (= one workgroup) decodes one independent block
// (reference: zxc_decompress_chunk_wrapper, src/lib/zxc_decompress.c:1646-1695;
// GLO body :847-1209, GHI body :1231-1469). Nothing here is a translation of the
// CPU loop: the CPU walks sequences one at a time with wild 16/32-byte copies;
// this kernel
//   1. parses 64 sequences per step, one per lane: token nibbles / GHI words,
//      varint escapes located with ballot + prefix popcount, varint boundaries
//      found by a wave-wide scan of 3-state transition maps (the prefix varint is
//      a 3-state automaton), output/literal cursors by wave prefix sums;
//   2. turns the batch into output bytes with one 16-byte chunk per lane
//      (binary search of the chunk start in the scanned sequence ends held in
//      LDS), so the 1 KiB a wave finishes per pass leaves as one coalesced
//      global_store_dwordx4 per lane;
//   3. keeps the last 16 KiB of output in an LDS ring (the sliding window);
//      back-references inside the ring are LDS byte-gathers (aligned dword reads +
//      v_alignbyte), older ones are L2 reads of the block's own output;
//   4. resolves matches that point into the pass being produced with a
//      ballot "final mask": a lane stalls until the lanes owning its source
//      chunks have published, no workgroup barrier involved. Overlapping matches
//      (offset < length) are rewritten to their period so a long run depends
//      only on the bytes in front of it.
// Integer byte work: no MFMA. Bounds: HBM (compressed in + decoded out).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zxc_dev.h"

typedef unsigned __int128 u128;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

#define RING_BYTES 16384u
#define RING_MASK (RING_BYTES - 1u)
#define RING_WORDS (RING_BYTES / 4u)

// zxc_error_t values (reference include/zxc_error.h:38-74)
#define E_DST_TOO_SMALL (-2)
#define E_SRC_TOO_SMALL (-3)
#define E_BAD_HEADER (-6)
#define E_CORRUPT (-8)
#define E_BAD_OFFSET (-9)
#define E_OVERFLOW (-10)
#define E_BAD_BLOCK_TYPE (-13)
#define E_DICT_REQUIRED (-15)

struct SeqRec {      // one sequence of the current 64-sequence batch
    uint32_t E;      // output position one past its match
    uint32_t M;      // output position where its match starts (= end of its literals)
    uint32_t off;    // match distance (>= 1)
    uint32_t lit;    // index of its first literal in the literal stream
};

struct __attribute__((aligned(16))) WaveLds {
    uint32_t ring[RING_WORDS + 8];  // +32 B mirror of the first 32 B: unaligned reads never wrap
    SeqRec seq[64];
    uint32_t vval[128];             // values of the batch's varints, in stream order
    uint32_t misc[4];
};

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ uint32_t ld8(const uint8_t* p) { return *p; }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ u128 ld128(const uint8_t* p) { u128 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ u128 mk128(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return (u128)a | ((u128)b << 32) | ((u128)c << 64) | ((u128)d << 96);
}

// wave-wide inclusive prefix sum (64 lanes)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t t = __shfl_xor(v, d);
        v = t < v ? t : v;
    }
    return v;
}
// LDS traffic between lanes of the one wave: order it, no s_barrier needed
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// bytes [b, b+n) of acc := bytes [0, n) of data   (b + n <= 16)
__device__ __forceinline__ u128 put_bytes(u128 acc, u128 data, uint32_t b, uint32_t n) {
    const u128 m = (n >= 16u) ? ~(u128)0 : ((((u128)1) << (8u * n)) - 1u);
    return (acc & ~(m << (8u * b))) | ((data & m) << (8u * b));
}

// 16 bytes of the sliding window starting at output position q (any alignment)
__device__ __forceinline__ u128 ring_fetch(const uint32_t* ring, uint32_t q) {
    const uint32_t i = (q & RING_MASK) >> 2;
    const uint32_t sh = q & 3u;
    const uint32_t w0 = ring[i], w1 = ring[i + 1], w2 = ring[i + 2], w3 = ring[i + 3], w4 = ring[i + 4];
    return mk128(__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
                 __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh));
}
__device__ __forceinline__ u128 ring_read_chunk(const uint32_t* ring, uint32_t cs) {
    const v4u v = *(const v4u*)(ring + ((cs & RING_MASK) >> 2));
    return mk128(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void ring_write_chunk(uint32_t* ring, uint32_t cs, u128 a) {
    const uint32_t i = (cs & RING_MASK) >> 2;
    v4u v;
    v.x = (uint32_t)a; v.y = (uint32_t)(a >> 32); v.z = (uint32_t)(a >> 64); v.w = (uint32_t)(a >> 96);
    *(v4u*)(ring + i) = v;
    if (i < 8u) *(v4u*)(ring + RING_WORDS + i) = v;
}


// 16 bytes of already-final output starting at q: from the LDS ring when recent
// enough, otherwise from the block's own output in global memory (nt load: served
// by L2, where this wave's earlier write-through stores already are).
__device__ __forceinline__ u128 fetch16(const uint32_t* ring, const uint8_t* dst, uint32_t q, uint32_t ring_lo,
                                        uint32_t out_pad) {
    if (q >= ring_lo) return ring_fetch(ring, q);
    if (q + 20u > out_pad) return 0;  // only reachable when a malformed block outgrows its slot
    const uint8_t* a = dst + (q & ~3u);
    const v4u g = __builtin_nontemporal_load((const v4u*)a);
    const uint32_t g4 = __builtin_nontemporal_load((const uint32_t*)(a + 16));
    const uint32_t sh = q & 3u;
    return mk128(__builtin_amdgcn_alignbyte(g.y, g.x, sh), __builtin_amdgcn_alignbyte(g.z, g.y, sh),
                 __builtin_amdgcn_alignbyte(g.w, g.z, sh), __builtin_amdgcn_alignbyte(g4, g.w, sh));
}

// x mod d for x < 2^22, 1 <= d <= 65536 (period rewrite of overlapping matches)
__device__ __forceinline__ uint32_t umod(uint32_t x, uint32_t d) {
    uint32_t q = (uint32_t)((float)x * __frcp_rn((float)d));
    int32_t r = (int32_t)(x - q * d);
    if (r < 0) r += (int32_t)d;
    if (r >= (int32_t)d) r -= (int32_t)d;
    return (uint32_t)r;
}

// ------------------------------------------------------------- varint batch parse
// The extras stream is a chain of 1..3-byte prefix varints (reference
// zxc_read_varint, src/lib/zxc_decompress.c:51-88). Each lane looks at 8 bytes
// of a 512-byte window starting at the cursor; "where does the first varint of my
// 8 bytes start" has 3 possible answers, so each lane's bytes are a map
// {0,1,2}->{0,1,2}; an inclusive wave scan of map composition gives every lane
// its true entry state. A bad (>= 0xE0) or truncated varint yields 0 and kills
// the stream (every later varint reads 0), exactly like the reference.
__device__ __forceinline__ uint32_t compose_map(uint32_t hi, uint32_t lo) {  // hi after lo
    uint32_t r = 0;
#pragma unroll
    for (int e = 0; e < 3; e++) r |= ((hi >> (2u * ((lo >> (2 * e)) & 3u))) & 3u) << (2 * e);
    return r;
}

__device__ void parse_varints(const uint8_t* ext, uint32_t ext_size, uint32_t& cur, uint32_t& dead,
                              uint32_t nv, WaveLds& L, int lane) {
    const uint32_t base = cur + 8u * (uint32_t)lane;
    uint64_t lo8;
    uint32_t hi4;
    if (base + 12u <= ext_size) {
        lo8 = ld64(ext + base);
        hi4 = ld32(ext + base + 8);
    } else {
        lo8 = 0;
        hi4 = 0;
        for (uint32_t k = 0; k < 11u; k++) {
            const uint32_t b = (base + k < ext_size) ? ld8(ext + base + k) : 0xFFu;
            if (k < 8u) lo8 |= (uint64_t)b << (8u * k);
            else hi4 |= b << (8u * (k - 8u));
        }
    }
    uint32_t len[8];
    uint32_t badbits = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t b = (uint32_t)(lo8 >> (8 * j)) & 255u;
        len[j] = 1u + (b >= 0x80u) + (b >= 0xC0u);
        if (b >= 0xE0u || base + j + len[j] > ext_size) badbits |= 1u << j;
    }
    uint32_t c0 = 0, c1 = 1, c2 = 2;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (c0 == (uint32_t)j) c0 += len[j];
        if (c1 == (uint32_t)j) c1 += len[j];
        if (c2 == (uint32_t)j) c2 += len[j];
    }
    uint32_t F = (c0 - 8u) | ((c1 - 8u) << 2) | ((c2 - 8u) << 4);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t G = __shfl_up(F, d);
        if (lane >= d) F = compose_map(F, G);
    }
    const uint32_t prevF = __shfl_up(F, 1);
    uint32_t c = (lane == 0) ? 0u : (prevF & 3u);
    uint32_t starts = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (c == (uint32_t)j) {
            starts |= 1u << j;
            c += len[j];
        }
    }
    const uint32_t cnt = __popc(starts);
    const uint32_t rank0 = wave_scan_add(cnt, lane) - cnt;
    const u128 W = (u128)lo8 | ((u128)hi4 << 64);
    uint32_t minbad = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (starts & (1u << j)) {
            const uint32_t k = rank0 + __popc(starts & ((1u << j) - 1u));
            if (k < nv) {
                const uint32_t v3 = (uint32_t)(W >> (8 * j)) & 0xFFFFFFu;
                const uint32_t b0 = v3 & 255u, b1 = (v3 >> 8) & 255u, b2 = v3 >> 16;
                uint32_t val = (b0 < 0x80u) ? b0 : (b0 < 0xC0u) ? ((b0 & 0x3Fu) | (b1 << 6))
                                                                  : ((b0 & 0x1Fu) | (b1 << 5) | (b2 << 13));
                if (badbits & (1u << j)) {
                    val = 0;
                    minbad = k < minbad ? k : minbad;
                }
                L.vval[k] = val;
                if (k == nv - 1u) L.misc[0] = base + j + len[j];
            }
        }
    }
    const uint32_t kbad = (__ballot(minbad != 0xFFFFFFFFu) != 0ull) ? wave_min(minbad) : 0xFFFFFFFFu;
    wave_lds_fence();
    if (kbad != 0xFFFFFFFFu) {
        // everything from the bad varint on reads as 0 (cursor parked at the end)
        if ((uint32_t)lane + kbad < nv) L.vval[lane + kbad] = 0;
        if ((uint32_t)lane + 64u + kbad < nv) L.vval[lane + 64 + kbad] = 0;
        dead = 1;
        cur = ext_size;
        wave_lds_fence();
    } else {
        cur = uni(L.misc[0]);
    }
}

// ------------------------------------------------------------------ block decode
struct LzStreams {
    const uint8_t* lit;   // literal bytes (payload, or expanded scratch)
    uint32_t n_lit;
    const uint8_t* tok;   // GLO: 1 B/seq tokens. GHI: 4 B/seq words
    const uint8_t* offs;  // GLO only
    const uint8_t* ext;
    uint32_t ext_size;
    uint32_t n_seq;
    uint32_t off8;        // GLO 1-byte offsets
    uint32_t ghi;
};

// Executes all sequences of one block. Returns decoded size or a negative error.
// dst must be 16-byte aligned; only bytes below out_len are stored.
__device__ int run_sequences(const LzStreams& S, uint8_t* __restrict__ dst, uint32_t out_len, uint32_t cap,
                             WaveLds& L, int lane) {
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t n_total = S.n_seq + 1u;  // + pseudo sequence carrying the trailing literals
    const uint32_t out_pad = (out_len + 15u) & ~15u;
    uint32_t p = 0, lp = 0, cur = 0, dead = 0;

    for (uint32_t seq_base = 0; seq_base < n_total; seq_base += 64u) {
        const uint32_t s = seq_base + (uint32_t)lane;
        const bool real = s < S.n_seq;
        uint32_t ll = 0, ml = 0, off = 1;
        bool escL = false, escM = false;
        if (real) {
            if (S.ghi) {
                const uint32_t w = ld32(S.tok + 4ull * s);
                ll = w >> 24;
                ml = (w >> 16) & 255u;
                off = (w & 0xFFFFu) + 1u;
                escL = ll == 255u;
                escM = ml == 255u;
            } else {
                const uint32_t t = ld8(S.tok + s);
                ll = t >> 4;
                ml = t & 15u;
                off = 1u + (S.off8 ? ld8(S.offs + s) : ld16(S.offs + 2ull * s));
                escL = ll == 15u;
                escM = ml == 15u;
            }
        }
        const uint64_t mL = __ballot(escL), mM = __ballot(escM);
        const uint32_t nv = __popcll(mL) + __popcll(mM);
        if (nv != 0u) {
            if (!dead) {
                parse_varints(S.ext, S.ext_size, cur, dead, nv, L, lane);
                const uint32_t r = __popcll(mL & lt_mask) + __popcll(mM & lt_mask);
                if (escL) ll += L.vval[r];
                if (escM) ml += L.vval[r + (escL ? 1u : 0u)];
                wave_lds_fence();
            }
        }
        if (real) ml += 5u;

        // cursors: inclusive scans of (ll+ml) and ll
        const uint32_t len = ll + ml;
        const uint32_t Eincl = wave_scan_add(len, lane);
        const uint32_t Lincl = wave_scan_add(ll, lane);
        const uint32_t est = p + (Eincl - len);    // where this sequence's literals land
        const uint32_t lst = lp + (Lincl - ll);    // its first literal
        int err = 0;
        if (real) {
            if (est > cap || len > cap - est || lst > S.n_lit || ll > S.n_lit - lst) err = E_OVERFLOW;
            else if (off > est + ll) err = E_BAD_OFFSET;
        } else if (s == S.n_seq) {
            if (est > cap || lst > S.n_lit || S.n_lit - lst > cap - est) err = E_OVERFLOW;
            else ll = S.n_lit - lst;  // trailing literals
        }
        const uint64_t em = __ballot(err != 0);
        if (em) return __shfl(err, __ffsll((unsigned long long)em) - 1);

        SeqRec r;
        r.M = est + ll;
        r.E = (s == S.n_seq) ? r.M : est + len;
        r.off = off;
        r.lit = lst;
        if (s > S.n_seq) r.E = r.M = 0xFFFFFFFFu;
        L.seq[lane] = r;
        const uint32_t last = (n_total - seq_base > 64u) ? 63u : (n_total - seq_base - 1u);
        const uint32_t tile_end = __shfl(r.E, last);
        const uint32_t lit_end = __shfl(lst + ll, last);
        wave_lds_fence();

        // ---- produce [p, tile_end): 16 B per lane, 1 KiB per pass
        const uint32_t c_end = (tile_end + 15u) >> 4;
        for (uint32_t c0 = p >> 4; c0 < c_end; c0 += 64u) {
            const uint32_t cs = (c0 + (uint32_t)lane) << 4;
            const uint32_t lo = cs > p ? cs : p;
            const uint32_t hi = (cs + 16u < tile_end) ? cs + 16u : tile_end;
            const bool active = lo < hi;
            const uint32_t span_base = c0 << 4;
            const uint32_t span_end = span_base + 1024u;
            const uint32_t ring_lo = span_end > RING_BYTES ? span_end - RING_BYTES : 0u;

            uint32_t j = 0, sstart = p, pos = lo;
            u128 acc = 0;
            SeqRec rec = {0, 0, 1, 0};
            if (active) {
#pragma unroll
                for (int st = 32; st >= 1; st >>= 1)
                    if (L.seq[j + st - 1].E <= lo) j += st;
                if (j > 0) sstart = L.seq[j - 1].E;
                rec = L.seq[j];
                if (lo > cs) acc = ring_read_chunk(L.ring, cs);  // bytes the previous batch left in this chunk
            }
            bool done = !active;
            for (uint32_t round = 0;; round++) {
                const uint64_t fm = __ballot(done);
                if (fm == ~0ull) break;
                if (round > 80u) return ZXC_DEV_E_INTERNAL;  // cannot happen: the lowest pending lane always finishes
                if (!done) {
                    bool stall = false;
                    uint32_t guard = 0;
                    while (!stall && pos < hi && ++guard < 64u) {
                        if (pos < rec.M) {  // literal run
                            const uint32_t lim = rec.M < hi ? rec.M : hi;
                            const uint32_t n = lim - pos;
                            acc = put_bytes(acc, ld128(S.lit + rec.lit + (pos - sstart)), pos - cs, n);
                            pos += n;
                        } else if (pos < rec.E) {  // match
                            const uint32_t lim = rec.E < hi ? rec.E : hi;
                            uint32_t n = lim - pos;
                            const uint32_t off_ = rec.off;
                            const uint32_t within = pos - rec.M;
                            if (off_ >= 16u) {
                                // source never touches this lane's own chunk. If the plain source
                                // would fall inside the match itself, fold it onto the period.
                                uint32_t q = pos - off_, n1 = n;
                                if (within >= off_) {
                                    const uint32_t rr = umod(within, off_);
                                    q = rec.M - off_ + rr;
                                    n1 = (off_ - rr < n) ? off_ - rr : n;
                                }
                                const uint32_t qe = q + n1;
                                bool ready = true;
                                if (qe > span_base) {
                                    const uint32_t a = (q > span_base ? q - span_base : 0u) >> 4;
                                    const uint32_t b = (qe - 1u - span_base) >> 4;
                                    ready = ((fm >> a) & (fm >> b) & 1ull) != 0ull;
                                }
                                if (!ready) { stall = true; break; }
                                const u128 d = fetch16(L.ring, dst, q, ring_lo, out_pad);
                                acc = put_bytes(acc, d, pos - cs, n1);
                                pos += n1;  // a folded copy may leave n - n1 bytes for the next turn
                            } else {
                                // short period: byte loop; sources are [M-off, M) only
                                const uint32_t B = rec.M - off_;
                                const uint32_t pe = rec.M < cs ? rec.M : cs;  // pattern bytes below pe live outside this lane
                                bool ready = true;
                                if (B < pe && pe > span_base) {  // some of them are produced in this very pass
                                    const uint32_t a = (B > span_base ? B - span_base : 0u) >> 4;
                                    const uint32_t b = (pe - 1u - span_base) >> 4;
                                    ready = ((fm >> a) & (fm >> b) & 1ull) != 0ull;
                                }
                                if (!ready) { stall = true; break; }
                                u128 pat = 0;
                                if (B < cs) pat = fetch16(L.ring, dst, B, ring_lo, out_pad);
                                uint32_t rr = within < off_ ? within : umod(within, off_);
                                for (uint32_t k = 0; k < n; k++) {
                                    const uint32_t sp = B + rr;
                                    const uint32_t byte = (sp >= cs) ? (uint32_t)(acc >> (8u * (sp - cs))) & 255u
                                                                     : (uint32_t)(pat >> (8u * rr)) & 255u;
                                    acc = put_bytes(acc, (u128)byte, pos + k - cs, 1u);
                                    rr = (rr + 1u == off_) ? 0u : rr + 1u;
                                }
                                pos += n;
                            }
                        }
                        if (pos >= rec.E && pos < hi) {
                            sstart = rec.E;
                            j++;
                            rec = L.seq[j & 63u];
                        }
                    }
                    if (pos >= hi) {
                        ring_write_chunk(L.ring, cs, acc);
                        if (hi == cs + 16u) {  // chunk complete: one coalesced 16 B store per lane
                            if (cs + 16u <= out_len) {
                                v4u v;
                                v.x = (uint32_t)acc; v.y = (uint32_t)(acc >> 32);
                                v.z = (uint32_t)(acc >> 64); v.w = (uint32_t)(acc >> 96);
                                *(v4u*)(dst + cs) = v;
                            } else {
                                for (uint32_t k = 0; cs + k < out_len && k < 16u; k++)
                                    dst[cs + k] = (uint8_t)(acc >> (8u * k));
                            }
                        }
                        done = true;
                    }
                }
                wave_lds_fence();
            }
        }
        p = tile_end;
        lp = lit_end;
        wave_lds_fence();
    }
    // the last chunk may be partial: it only lives in the ring so far
    if ((p & 15u) != 0u && lane == 0) {
        const uint32_t cs = p & ~15u;
        const u128 a = ring_read_chunk(L.ring, cs);
        for (uint32_t k = 0; cs + k < p && cs + k < out_len; k++) dst[cs + k] = (uint8_t)(a >> (8u * k));
    }
    return (int)p;
}

// RLE literal section -> scratch (reference src/lib/zxc_decompress.c:906-975).
// v1: one lane walks the tokens. TODO(perf): wave-parallel token chain.
__device__ int rle_expand(const uint8_t* r, uint32_t rsize, uint8_t* w, uint32_t n, int lane) {
    int rc = 0;
    if (lane == 0) {
        uint32_t ri = 0, wi = 0;
        while (ri < rsize && wi < n) {
            const uint32_t tok = r[ri++];
            if (!(tok & 0x80u)) {
                const uint32_t len = tok + 1u;
                if (n - wi < len || rsize - ri < len) { rc = E_CORRUPT; break; }
                for (uint32_t k = 0; k < len; k++) w[wi + k] = r[ri + k];
                wi += len;
                ri += len;
            } else {
                const uint32_t len = (tok & 0x7Fu) + 4u;
                if (n - wi < len || ri >= rsize) { rc = E_CORRUPT; break; }
                const uint8_t v = r[ri++];
                for (uint32_t k = 0; k < len; k++) w[wi + k] = v;
                wi += len;
            }
        }
        if (rc == 0 && wi != n) rc = E_CORRUPT;
    }
    rc = __shfl(rc, 0);
    // make lane 0's stores visible to every lane's (L1-cached) loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return rc;
}

__device__ int decode_lz_block(const uint8_t* data, uint32_t comp_sz, bool ghi, uint8_t* dst, uint32_t out_len,
                               uint32_t cap, uint32_t block_size, uint8_t* scratch, WaveLds& L, int lane) {
    if (comp_sz < 12u) return E_BAD_HEADER;
    LzStreams S;
    S.n_seq = uni(ld32(data));
    S.n_lit = uni(ld32(data + 4));
    const uint32_t enc_lit = uni(ld8(data + 8)), enc_tok = uni(ld8(data + 9)), enc_off = uni(ld8(data + 11));
    S.ghi = ghi;
    if (ghi) {
        if (enc_lit != 0u || enc_tok != 0u) return E_CORRUPT;
        const uint32_t avail = comp_sz - 12u;
        const uint64_t consumed = (uint64_t)S.n_lit + 4ull * S.n_seq;
        if (consumed > avail || avail - S.n_lit < 32u) return E_CORRUPT;
        S.lit = data + 12;
        S.tok = S.lit + S.n_lit;
        S.offs = nullptr;
        S.off8 = 0;
        S.ext = S.tok + 4ull * S.n_seq;
        S.ext_size = avail - (uint32_t)consumed;
        return run_sequences(S, dst, out_len, cap, L, lane);
    }
    const uint32_t desc = (enc_lit != 0u ? 4u : 0u) + (enc_tok == 2u ? 4u : 0u);
    if (comp_sz < 12u + desc) return E_BAD_HEADER;
    uint32_t lit_comp = S.n_lit, tok_comp = S.n_seq;
    const uint8_t* d = data + 12;
    if (enc_lit != 0u) { lit_comp = uni(ld32(d)); d += 4; }
    if (enc_tok == 2u) { tok_comp = uni(ld32(d)); d += 4; }
    if (enc_off > 1u) return E_CORRUPT;
    const uint8_t* pdata = data + 12 + desc;
    const uint32_t avail = comp_sz - 12u - desc;
    S.lit = pdata;
    if (enc_lit == 2u || enc_lit == 3u) {
        if (lit_comp > avail) return E_CORRUPT;
        if (S.n_lit != 0u) {
            if (S.n_lit > cap) return E_DST_TOO_SMALL;
            if (enc_lit == 3u) return E_DICT_REQUIRED;
            return ZXC_DEV_E_UNSUPPORTED;  // PivCo literal section: not in this kernel yet
        }
    } else if (enc_lit == 1u) {
        if (S.n_lit != 0u) {
            if (S.n_lit > cap) return E_DST_TOO_SMALL;
            if (S.n_lit > block_size || lit_comp > avail) return E_CORRUPT;
            const int rc = rle_expand(pdata, lit_comp, scratch, S.n_lit, lane);
            if (rc != 0) return rc;
            S.lit = scratch;
        }
    } else if (enc_lit != 0u) {
        return E_CORRUPT;
    }
    const uint64_t sz_off = (uint64_t)S.n_seq * (enc_off ? 1u : 2u);
    const uint64_t consumed = (uint64_t)lit_comp + tok_comp + sz_off;
    if (consumed > avail || avail - lit_comp < 32u) return E_CORRUPT;
    if (enc_tok == 2u) return ZXC_DEV_E_UNSUPPORTED;  // PivCo token section
    if (enc_tok != 0u) return E_CORRUPT;
    S.tok = pdata + lit_comp;
    S.offs = S.tok + tok_comp;
    S.off8 = enc_off;
    S.ext = S.offs + sz_off;
    S.ext_size = avail - (uint32_t)consumed;
    return run_sequences(S, dst, out_len, cap, L, lane);
}

extern "C" __global__ void __launch_bounds__(64)
zxc_decode_blocks_kernel(const uint8_t* __restrict__ comp, const zxc_dev_job_t* __restrict__ jobs, uint32_t n_jobs,
                         uint8_t* __restrict__ out, int32_t* __restrict__ status, uint32_t block_size,
                         uint32_t trailer_bytes, uint8_t* __restrict__ scratch, uint32_t scratch_stride) {
    __shared__ WaveLds L;
    const int lane = threadIdx.x;
    const uint32_t cap = block_size + 2112u;  // the reference always decodes with block_size + ZXC_DECOMPRESS_TAIL_PAD
    uint8_t* my_scratch = scratch + (size_t)blockIdx.x * scratch_stride;
    for (uint32_t b = blockIdx.x; b < n_jobs; b += gridDim.x) {
        const uint64_t comp_off = jobs[b].comp_off;
        const uint32_t src_sz = uni(jobs[b].comp_size);
        const uint32_t out_len = uni(jobs[b].out_len);
        const uint8_t* src = comp + comp_off;
        uint8_t* dst = out + jobs[b].out_off;
        int rc;
        if (src_sz < 8u) {
            rc = E_SRC_TOO_SMALL;
        } else {
            const uint32_t type = uni(ld8(src));
            const uint32_t comp_sz = uni(ld32(src + 3));
            if ((uint64_t)8u + comp_sz + trailer_bytes > src_sz) {
                rc = E_SRC_TOO_SMALL;
            } else if (type == 1u || type == 2u) {
                rc = decode_lz_block(src + 8, comp_sz, type == 2u, dst, out_len, cap, block_size, my_scratch, L, lane);
            } else if (type == 0u) {  // RAW: stored bytes
                if (comp_sz > cap) rc = E_DST_TOO_SMALL;
                else {
                    const uint32_t n = comp_sz < out_len ? comp_sz : out_len;
                    const uint8_t* s8 = src + 8;
                    for (uint32_t i = 16u * lane; i + 16u <= n; i += 1024u) {
                        const u128 v = ld128(s8 + i);
                        v4u w;
                        w.x = (uint32_t)v; w.y = (uint32_t)(v >> 32); w.z = (uint32_t)(v >> 64); w.w = (uint32_t)(v >> 96);
                        *(v4u*)(dst + i) = w;
                    }
                    const uint32_t tail = n & ~15u;
                    if (tail + (uint32_t)lane < n) dst[tail + lane] = s8[tail + lane];
                    rc = (int)comp_sz;
                }
            } else if (type == 255u) {
                rc = E_CORRUPT;
            } else {
                rc = E_BAD_BLOCK_TYPE;
            }
        }
        if (lane == 0) status[b] = rc;
    }
}
