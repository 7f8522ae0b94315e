// zxc_decode_kernel.hip — ZXC v8 block decode for gfx950 (MI355X, CDNA4).
//
// One 64-lane wavefront (= one workgroup) decodes one independent block
// (reference: zxc_decompress_chunk_wrapper, src/lib/zxc_decompress.c:1646-1695;
// GLO body :847-1209, GHI body :1231-1469). Nothing here is a translation of the
// CPU loop (one sequence at a time, wild 16/32-byte copies). Per block the wave
//   1. parses 64 sequences per batch, one per lane: token nibbles / GHI words and the window of
//      the extras stream were requested during the previous batch; varint escapes located with
//      ballot + prefix popcount, single-byte varints read cross-lane from the window, longer ones
//      by a wave-wide scan of 3-state transition maps (the prefix varint is a 3-state automaton);
//      output / literal cursors by DPP prefix sums; all bounds checks branch-free;
//   2. keeps the last RING_BYTES (4 KiB) of output — the sliding window — in an LDS ring whose
//      not-yet-written part is kept ZERO. A literal run or a match of any alignment is then put
//      as whole dwords: the bytes are shifted onto the destination's dword grid once, masked to
//      the run, and OR-ed in with ds_or_b32 (two neighbouring runs meet in one dword without a
//      read-modify-write in registers and without byte stores). Literals arrive from memory already
//      on the destination grid (the load address is biased by the destination's byte phase),
//      ring sources need one v_alignbyte per dword, sources older than the ring come back from
//      the block's own output through L2 with the same biased unaligned load;
//   3. orders matches that depend on other matches of the same batch with exact dependency masks
//      checked against a ballot of finished lanes (no barrier); simple containments are redirected
//      to the earlier match's own source; the last few dependent ones are finished one by one, in
//      stream order, by the whole wave;
//   4. long copies are done by the whole wave, 16 B per lane; overlapping matches (offset <
//      length) become a series of non-overlapping copies whose distance doubles (a period stays
//      a period), so runs need no byte loop;
//   5. streams finished 16-byte chunks from the ring to HBM, one coalesced
//      global_store_dwordx4 per lane, and zeroes the part of the ring the next batch will fill.
// RLE and PivCo (Huffman) literal / token sections are expanded first into a scratch slot
// (rle_expand below, zxc_pivco.inc); checksums by zxc_rapidhash.inc; a dictionary prefix is a
// template variant. Launch order: heaviest blocks first (zxc_order_* kernels at the end).
// Integer byte work: no MFMA. Bound: HBM (compressed bytes in + decoded bytes out).
#include "zxc_experiments.h"  // (first: the gate in front of every experiment switch)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zxc_dev.h"

typedef unsigned __int128 u128;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef v4u __attribute__((aligned(1))) v4u_unaligned;

#include "zxc_lds.h"

#ifndef RING_BYTES
#define RING_BYTES 4096u   // sliding window kept in LDS; older history is read back through L2
#endif
#define RING_MASK (RING_BYTES - 1u)
#define RING_WORDS (RING_BYTES / 4u)
#ifndef REDIRECT_PASSES
#define REDIRECT_PASSES 2   // chained containment levels resolved before round 0 (A/B: 0: -6 %, 1: -2 %, 2: best, 3: -1 %)
#endif
#ifndef SPARSE_MAX
#define SPARSE_MAX 8   // at most this many unfinished sequences after a round: finish them one by one (A/B: 2: -0.3 %, 4: 0, 8: +1 %, 12 / 16: 0)
#endif
#ifndef TILE_MAX
// A batch never spans more output than this: the ring must hold the batch, the not yet flushed tail of the
// previous one (< 16 B) and the round-up of the zeroed region to 16 B.
#define TILE_MAX 3584u
#endif
#ifndef LIT_MED
#define LIT_MED 128u     // literal runs up to this long are put lane-per-sequence (16-byte grid steps), longer ones by the whole wave
                         // (A/B on the bench corpus: 32: -9 %, 48: 0, 96: +8 %, 128 / 160 / 224: +10 %)
#endif
#ifndef MATCH_MED
#define MATCH_MED 128u   // matches up to this long are copied lane-per-sequence (16-byte grid steps)
#endif
#ifndef LIT_LOOP2
#define LIT_LOOP2 1      // literal groups beyond the third: two per step, requested unconditionally (0: one conditional load per step)
#endif
#ifndef LIT_PRELOAD
#define LIT_PRELOAD 3    // literal groups of a run requested before the dependency analysis (2: the third joins the loop; register diet A/B)
#endif
#ifndef FAR_PRELOAD
#define FAR_PRELOAD 2    // far-source groups of a match requested before the literal puts (1: the second is fetched in its step)
#endif
#ifndef FAR_EARLY
#define FAR_EARLY 0      // far-source groups: 0 requested after the literal wait; 1 by every lane, unconditionally, before it (A/B: -6.5 %);
                         // 2 by the lanes that need them, before the dependency analysis (A/B: -4.5 %)
#endif
#define BYTEWISE_MAX 32u // short-period overlapping matches up to this long: lane-local byte loop

// zxc_error_t values (reference include/zxc_error.h:38-74)
#define E_DST_TOO_SMALL (-2)
#define E_SRC_TOO_SMALL (-3)
#define E_BAD_HEADER (-6)
#define E_BAD_CHECKSUM (-7)
#define E_CORRUPT (-8)
#define E_BAD_OFFSET (-9)
#define E_OVERFLOW (-10)
#define E_BAD_BLOCK_TYPE (-13)
#define E_DICT_REQUIRED (-15)

// timing-ablation switches of the PivCo decoder (experiment builds only; see ZXC_EXPERIMENT below)
#define DBG_NO_SEQ 512u      // stop after the literal / token sections are expanded (timing only)
#define DBG_PIV_NO_P2 1024u  // PivCo: skip the bottom-up merges (timing only)
#define DBG_PIV_NO_P1 2048u  // PivCo: skip everything after the tree set-up (timing only)

#ifdef EXP_PHASES  // experiment only: per-phase shader-clock totals of each block, written over the block's first 32 output bytes
#define PH(i) do { const uint64_t t_ = __builtin_readcyclecounter(); ph[i] += (uint32_t)(t_ - ph_last); ph_last = t_; } while (0)
#define PHC(i) do { ph[i] += 1u; } while (0)
#else
#define PH(i) do { } while (0)
#define PHC(i) do { } while (0)
#endif

struct __attribute__((aligned(16))) RingLds {
    uint32_t ring[RING_WORDS];          // last RING_BYTES of output, position p at byte p & RING_MASK; bytes not yet written are zero
    uint32_t tmask[17 * 4];             // tmask[4 t + j]: byte mask of dword j of a 16-byte group that keeps the group's first t bytes
};
struct __attribute__((aligned(16))) WaveLds : RingLds {
    uint32_t vval[128];                 // values of the batch's varints, in stream order
    uint32_t vpos[130];                 // their byte positions in the extras stream (+ end cursor)
};

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ uint32_t ld8(const uint8_t* p) { return *p; }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ v4u ld128(const uint8_t* p) { v4u v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// Arguments of an out-of-line device function arrive in VGPRs and count as divergent; this tells
// the compiler a pointer is wave-uniform again (scalar address arithmetic and branches).
template <typename T>
__device__ __forceinline__ T* uni_ptr(T* p) {
    const uint64_t a = (uint64_t)p;
    return (T*)(((uint64_t)uni((uint32_t)(a >> 32)) << 32) | uni((uint32_t)a));
}

// wave-wide inclusive prefix sum (64 lanes) on the DPP crossbar: row_shr 1,2,4,8 scan each
// 16-lane row, row_bcast:15 / row_bcast:31 carry the row totals across (gfx9 DPP controls).
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v, int lane) {
    (void)lane;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
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
// LDS traffic between lanes of the one wave. One wave's DS instructions execute in program order, so a read issued
// after another lane's write sees it: only the COMPILER must keep the order (wavefront-scope fence: no instruction,
// where a workgroup-scope one drains the LDS queue with s_waitcnt lgkmcnt(0), ~10 times per batch).
#ifndef LDS_FENCE_SCOPE
#define LDS_FENCE_SCOPE "wavefront"
#endif
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, LDS_FENCE_SCOPE);
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------ ring access
__device__ __forceinline__ uint8_t* ring8(RingLds& L) { return (uint8_t*)L.ring; }
__device__ __forceinline__ uint32_t ring_rd8(RingLds& L, uint32_t pos) { return LDS_LD8(ring8(L) + (pos & RING_MASK)); }
__device__ __forceinline__ void ring_wr8(RingLds& L, uint32_t pos, uint32_t v) { LDS_ST8(ring8(L) + (pos & RING_MASK), v); }
__device__ __forceinline__ uint32_t ring_rd32(RingLds& L, uint32_t pos4) { return LDS_LD32(ring8(L) + (pos4 & RING_MASK)); }
__device__ __forceinline__ void ring_or32(RingLds& L, uint32_t pos4, uint32_t v) { LDS_OR32(ring8(L) + (pos4 & RING_MASK), v); }
// 16 bytes starting at any position: aligned dword reads + v_alignbyte
__device__ __forceinline__ v4u ring_rd128(RingLds& L, uint32_t pos) {
    const uint32_t b = pos & ~3u;
    const uint32_t sh = pos & 3u;
    const uint32_t w0 = ring_rd32(L, b), w1 = ring_rd32(L, b + 4u), w2 = ring_rd32(L, b + 8u), w3 = ring_rd32(L, b + 12u),
                   w4 = ring_rd32(L, b + 16u);
    v4u r;
    r.x = __builtin_amdgcn_alignbyte(w1, w0, sh);
    r.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
    r.z = __builtin_amdgcn_alignbyte(w3, w2, sh);
    r.w = __builtin_amdgcn_alignbyte(w4, w3, sh);
    return r;
}
__device__ __forceinline__ v4u ring_rd128_aligned(RingLds& L, uint32_t pos16) { return LDS_LD128(ring8(L) + (pos16 & RING_MASK)); }
__device__ __forceinline__ void ring_wr128_aligned(RingLds& L, uint32_t pos16, v4u v) { LDS_ST128(ring8(L) + (pos16 & RING_MASK), v); }

// A group = 4 dwords of the destination's dword grid (16 bytes at a 4-aligned position).
// Keeps the first t (0..16) bytes of a group.
__device__ __forceinline__ v4u group_keep_first(RingLds& L, const v4u d, uint32_t t) {
    const v4u m = LDS_LD128((const uint8_t*)L.tmask + 16u * t);
    return d & m;
}
// Byte mask that clears the first a (0..3) bytes of a dword: bytes [3-a, 7-a) of 00 00 00 FF FF FF FF FF.
__device__ __forceinline__ uint32_t head_mask(uint32_t a) { return __builtin_amdgcn_alignbyte(0xFFFFFFFFu, 0xFF000000u, 3u - a); }
__device__ __forceinline__ void ring_or_group(RingLds& L, uint32_t gpos4, const v4u d) {
    ring_or32(L, gpos4, d.x);
    ring_or32(L, gpos4 + 4u, d.y);
    ring_or32(L, gpos4 + 8u, d.z);
    ring_or32(L, gpos4 + 12u, d.w);
}

// Per-block view shared by the copy routines.
struct Out {
    uint8_t* dst;        // block's slot in the output buffer (16-byte aligned)
    uint32_t out_len;    // bytes of it the caller keeps
    uint32_t out_pad;    // out_len rounded up to 16 (readable/writable)
    uint32_t flushed;    // output positions below this are in global memory (multiple of 16)
    const uint8_t* dict; // dictionary prefix: logically occupies output positions [-dict_size, 0)
    uint32_t dict_size;
};

// Already-final output at position q that has left the ring: read it back from the
// block's own output (nt: served by L2, where the write-through stores already are).
// Global loads take any byte alignment, so the 16 bytes come back in place.
__device__ __forceinline__ v4u far_rd128(const Out& O, uint32_t q) {
    v4u z = {0, 0, 0, 0};
    if (q + 16u > O.out_pad) return z;  // only a malformed, oversize block
    return __builtin_nontemporal_load((const v4u_unaligned*)(O.dst + q));
}
__device__ __forceinline__ uint32_t far_rd8(const Out& O, uint32_t q) {
    if (q >= O.out_pad) return 0;
    return __builtin_nontemporal_load(O.dst + q);
}

// One source byte at output position s, where a negative s (as int32) addresses the dictionary
// prefix that logically precedes the block (reference: d_floor = dst - dict_size,
// src/lib/zxc_decompress.c:1028).
template <bool DICT>
__device__ __forceinline__ uint32_t src_rd8(RingLds& L, const Out& O, uint32_t s, uint32_t ring_lo) {
    if (DICT && (int32_t)s < 0) return ld8(O.dict + (int32_t)(O.dict_size + s));
    return s >= ring_lo ? ring_rd8(L, s) : far_rd8(O, s);
}

// Stream finished output from the ring to HBM: chunks [O.flushed, upto & ~15).
__device__ __forceinline__ void flush_to(RingLds& L, Out& O, uint32_t upto, int lane) {
    const uint32_t end = upto & ~15u;
    for (uint32_t c = O.flushed + 16u * (uint32_t)lane; c < end; c += 1024u) {
        const v4u v = ring_rd128_aligned(L, c);
        if (c + 16u <= O.out_len) {
#ifdef FLUSH_NT
            __builtin_nontemporal_store(v, (v4u*)(O.dst + c));
#else
            *(v4u*)(O.dst + c) = v;  // one coalesced 16 B store per lane
#endif
        } else if (c < O.out_len) {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (uint32_t k = 0; c + k < O.out_len; k++) O.dst[c + k] = (uint8_t)(w[k >> 2] >> (8u * (k & 3u)));
        }
    }
    if (end > O.flushed) O.flushed = end;
}

// Zero the ring for output positions [from16, to16) (both multiples of 16, at most RING_BYTES apart):
// the part of the window the coming batch fills with ds_or puts.
__device__ __forceinline__ void ring_zero(RingLds& L, uint32_t from16, uint32_t to16, int lane) {
    const v4u z = {0, 0, 0, 0};
    for (uint32_t c = from16 + 16u * (uint32_t)lane; c < to16; c += 1024u) ring_wr128_aligned(L, c, z);
}

// Whole-wave copy of n bytes to output position dpos. The source is either output
// position spos (ring when >= ring_lo, L2 otherwise; must not overlap the
// destination: n <= dpos - spos) or, when lit != nullptr, the literal stream.
// Plain (not OR) stores: every destination byte is written exactly once, whatever was there.
template <bool DICT>
__device__ void coop_copy(RingLds& L, const Out& O, uint32_t dpos, uint32_t spos, const uint8_t* lit, uint32_t n,
                          uint32_t ring_lo, int lane) {
    const uint32_t h0 = (16u - (dpos & 15u)) & 15u;
    const uint32_t h = h0 < n ? h0 : n;  // bytes up to the first 16-byte boundary
    if ((uint32_t)lane < h) {
        const uint32_t s = spos + lane;
        const uint32_t b = lit ? ld8(lit + lane) : src_rd8<DICT>(L, O, s, ring_lo);
        ring_wr8(L, dpos + lane, b);
    }
    const uint32_t nb = (n - h) >> 4;
    for (uint32_t c = lane; c < nb; c += 64u) {
        const uint32_t o = h + 16u * c;
        const uint32_t s = spos + o;
        v4u v;
        if (lit) v = ld128(lit + o);
        else if (DICT && (int32_t)s < 0) {  // (partly) inside the dictionary: byte gather (rare)
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint32_t k = 0; k < 16u; k++) w[k >> 2] |= src_rd8<DICT>(L, O, s + k, ring_lo) << (8u * (k & 3u));
            v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        } else if (s >= ring_lo) v = ring_rd128(L, s);
        else v = far_rd128(O, s);
        ring_wr128_aligned(L, dpos + o, v);
    }
    const uint32_t tl = (n - h) & 15u;
    if ((uint32_t)lane < tl) {
        const uint32_t o = n - tl + lane;
        const uint32_t s = spos + o;
        const uint32_t b = lit ? ld8(lit + o) : src_rd8<DICT>(L, O, s, ring_lo);
        ring_wr8(L, dpos + o, b);
    }
    wave_lds_fence();
}

// Whole-wave match copy [M, M+ml) <- [M-off, ...). An overlapping match (off < ml) is a
// period-off pattern: copy `dist` bytes from distance `dist`, then the valid periodic
// region has doubled, so double dist (it stays a multiple of off). With flush_each the
// ring is drained between steps (giant sequences, longer than the ring).
template <bool DICT>
__device__ void coop_match(RingLds& L, Out& O, uint32_t M, uint32_t ml, uint32_t off, uint32_t ring_lo_fixed,
                           bool flush_each, int lane) {
    uint32_t done = 0, dist = off;
    while (done < ml) {
        uint32_t n = ml - done;
        if (n > dist) n = dist;
        if (n > TILE_MAX) n = TILE_MAX;
        const uint32_t d = M + done;
        uint32_t ring_lo = ring_lo_fixed;
        if (flush_each) ring_lo = (d + n > RING_BYTES) ? d + n - RING_BYTES : 0u;
        coop_copy<DICT>(L, O, d, d - dist, nullptr, n, ring_lo, lane);
        done += n;
        if (flush_each) {
            flush_to(L, O, d + n, lane);
            __builtin_amdgcn_s_waitcnt(0);  // the next piece may read these bytes back from memory (distance > ring - piece)
        }
        if (n == dist && dist < TILE_MAX) dist <<= 1;
    }
}

// ------------------------------------------------------------- varint batch parse
// The extras stream is a chain of 1..3-byte prefix varints (reference
// zxc_read_varint, src/lib/zxc_decompress.c:51-88). Each lane looks at 8 bytes
// of a 512-byte window starting at the cursor; "where does the first varint of my
// 8 bytes start" has 3 possible answers, so each lane's bytes are a map
// {0,1,2}->{0,1,2}; an inclusive wave scan of map composition gives every lane
// its true entry state. Fills L.vval[k] / L.vpos[k] for k < nv, L.vpos[nv] = the
// cursor after them, and returns the rank of the first bad (>= 0xE0 or truncated)
// varint, which reads as 0 and kills the stream like in the reference.
__device__ __forceinline__ uint32_t compose_map(uint32_t hi, uint32_t lo) {  // hi after lo
    uint32_t r = 0;
#pragma unroll
    for (int e = 0; e < 3; e++) r |= ((hi >> (2u * ((lo >> (2 * e)) & 3u))) & 3u) << (2 * e);
    return r;
}

template <typename LDS>  // WaveLds (up to 128 varints per batch) or LeanLds (up to 62)
__device__ __forceinline__ uint32_t parse_varints(const uint8_t* ext, uint32_t ext_size, uint32_t cur, uint32_t nv, LDS& L,
                                  int lane) {
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
            if (k <= nv) L.vpos[k] = base + j;  // k == nv: where the next batch resumes
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
            }
        }
    }
    const uint32_t kbad = (__ballot(minbad != 0xFFFFFFFFu) != 0ull) ? wave_min(minbad) : 0xFFFFFFFFu;
    wave_lds_fence();
    return kbad;
}

// (Two other section decoders were built and measured this round — a top-down wavelet-tree walk and an LDS-tiled
// bottom-up merge, tools/experiments/zxc_pivco_*.inc — both slower on the GPU than this one: DESIGN.md §3.)
#ifdef PIV_VARIANT_FILE
#include PIV_VARIANT_FILE
#else
#include "zxc_pivco.inc"
#include "zxc_pivco_dir.inc"
#endif
#include "zxc_rapidhash.inc"

// ------------------------------------------------------------------ block decode
struct LzStreams {
    const uint8_t* lit;   // literal bytes (payload, or expanded scratch); 3 bytes before and 16 after are readable
    uint32_t n_lit;
    const uint8_t* tok;   // GLO: 1 B/seq tokens. GHI: 4 B/seq words
    const uint8_t* offs;  // GLO only
    const uint8_t* ext;
    uint32_t ext_size;
    uint32_t n_seq;
    uint32_t off8;        // GLO 1-byte offsets
    const uint8_t* dict;  // dictionary prefix (or nullptr)
    uint32_t dict_size;
};

// number of lanes whose (ascending over lanes) value is <= x: binary search by bpermute
// (A/B on the bench mix: fetching the first three levels' pivots with v_readlane and pairing the two searches'
// round trips, or counting the last eight lanes with seven independent bpermutes, is within +-1 %)
__device__ __forceinline__ uint32_t lanes_le(uint32_t sorted, uint32_t x) {
    uint32_t c = 0;
#pragma unroll
    for (int st = 32; st >= 1; st >>= 1) {
        const uint32_t t = __shfl(sorted, (int)(c + st - 1u));
        if (t <= x) c += st;
    }
    const uint32_t t63 = __shfl(sorted, 63);
    return c + ((c == 63u && t63 <= x) ? 1u : 0u);
}

// Token and offset of sequence s, undecoded (GHI: the 32-bit word; GLO: token byte, offset - 1). Requested by every lane,
// beyond the last sequence too (a conditional load would be waited for on the spot, and zero-initialising its register
// makes the compiler drain every older vector-memory access first): such lanes read the last sequence's bytes and the
// consumer masks them (n_seq >= 1 here, or the streams are empty and nothing is read).
template <bool GHI>
__device__ __forceinline__ void load_seq_raw(const LzStreams& S, uint32_t s, uint32_t& raw_t, uint32_t& raw_o) {
    raw_t = 0;
    raw_o = 0;
    if (S.n_seq == 0u) return;  // wave-uniform
    const uint32_t c = s < S.n_seq ? s : S.n_seq - 1u;
    if (GHI) raw_t = ld32(S.tok + 4ull * c);
    else {
        raw_t = ld8(S.tok + c);
        raw_o = S.off8 ? ld8(S.offs + c) : ld16(S.offs + 2ull * c);
    }
}

// Executes all sequences of one block. Returns decoded size or a negative error.
//
// The ring invariant that makes the ds_or puts work: at the top of every batch all ring bytes of
// output positions [p, z_end) are zero (p = bytes produced so far, z_end a multiple of 16), and
// before a batch is executed the zeroed region is extended to cover the batch ([z_end, z_new),
// stale window data one lap old that nothing can reference any more: ring_lo = z_new - RING_BYTES).
template <bool DICT, bool GHI>
__device__ __forceinline__ int run_sequences(const LzStreams& S, uint8_t* __restrict__ dst, uint32_t out_len, uint32_t cap,
                             WaveLds& L, int lane, const bool strict) {
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // the reference's 4x-batch reserve: see run_sequences_lean (zxc_seq_lean.inc)
    const uint32_t RW = GHI ? 513u : 33u, RD = GHI ? 2112u : 168u, RLM = GHI ? 1016u : 56u;
    bool carry4x = true;
    const uint32_t n_total = S.n_seq + 1u;  // + pseudo sequence carrying the trailing literals
    const uint32_t n_lit = S.n_lit;
    Out O;
    O.dst = dst;
    O.out_len = out_len;
    O.out_pad = (out_len + 15u) & ~15u;
    O.flushed = 0;
    O.dict = S.dict;
    O.dict_size = S.dict_size;
    uint32_t p = 0, lp = 0, cur = 0, dead = 0, seq_base = 0;
    uint32_t z_end = RING_BYTES;
#ifdef EXP_PHASES
    uint32_t ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t ph_last = __builtin_readcyclecounter();
#endif

    // tail-mask table + an all-zero ring
    for (uint32_t i = (uint32_t)lane; i < 68u; i += 64u) {
        const int v = (int)(i >> 2) - 4 * (int)(i & 3u);  // bytes of dword (i & 3) kept when the first (i >> 2) bytes of the group are
        LDS_ST32((uint8_t*)L.tmask + 4u * i, v <= 0 ? 0u : (v >= 4 ? 0xFFFFFFFFu : ((1u << (8 * v)) - 1u)));
    }
    ring_zero(L, 0u, RING_BYTES, lane);
    wave_lds_fence();

    uint32_t raw_t, raw_o;
    load_seq_raw<GHI>(S, (uint32_t)lane, raw_t, raw_o);
    // (requested by every lane, clamped to the stream's last byte — or the byte in front of an empty stream: only bytes
    // below the stream's end are ever looked at)
    const int32_t ext_last = (int32_t)S.ext_size - 1;
    uint32_t xw0 = ld8(S.ext + (lane < ext_last ? lane : ext_last));
    uint32_t xw1 = ld8(S.ext + (64 + lane < ext_last ? 64 + lane : ext_last));
    __builtin_amdgcn_s_waitcnt(0);  // (once per block: the loop is entered with nothing on the compiler's scoreboard, see the flush)

    while (seq_base < n_total) {
        const uint32_t s = seq_base + (uint32_t)lane;
        const bool real = s < S.n_seq;
        const bool valid = s < n_total;
        // (raw_t / raw_o were requested while the previous batch was being copied; beyond n_seq they hold the last sequence's)
        if (!real) { raw_t = 0; raw_o = 0; }
        uint32_t ll, ml, off;
        bool escL, escM;
        if (GHI) {
            ll = raw_t >> 24;
            ml = (raw_t >> 16) & 255u;
            off = (raw_t & 0xFFFFu) + 1u;
            escL = real & (ll == 255u);
            escM = real & (ml == 255u);
        } else {
            ll = raw_t >> 4;
            ml = raw_t & 15u;
            off = 1u + raw_o;
            escL = real & (ll == 15u);
            escM = real & (ml == 15u);
        }
        const uint64_t mL = __ballot(escL), mM = __ballot(escM);
        const uint32_t nv = __popcll(mL) + __popcll(mM);
        uint32_t kbad = 0xFFFFFFFFu;
        const bool parsed = nv != 0u && !dead;
        bool fast_vi = false;
        if (parsed) {
            const uint32_t r = __popcll(mL & lt_mask) + __popcll(mM & lt_mask);
            const uint32_t r2 = r + (escL ? 1u : 0u);
            // Fast path: all nv varints are single bytes (< 0x80), so the k-th sits at cur + k.
            uint32_t b0 = 0, b1 = 0;
            bool small = true;
            if (cur + nv <= S.ext_size) {
                // the 128 extras bytes at the cursor sit one per lane in xw0 / xw1 (requested during the
                // previous batch): rank -> byte is a cross-lane read, not a trip to memory
                const uint32_t x0 = __shfl(xw0, (int)(r & 63u)), y0 = __shfl(xw0, (int)(r2 & 63u));
                uint32_t x1 = 0, y1 = 0;
                if (nv > 64u) { x1 = __shfl(xw1, (int)(r & 63u)); y1 = __shfl(xw1, (int)(r2 & 63u)); }
                if (escL) { b0 = r < 64u ? x0 : x1; small = small && b0 < 0x80u; }
                if (escM) { b1 = r2 < 64u ? y0 : y1; small = small && b1 < 0x80u; }
                fast_vi = __ballot(!small) == 0ull;
            }
            if (fast_vi) {
                ll += b0;
                ml += b1;
            } else {
                kbad = parse_varints(S.ext, S.ext_size, cur, nv, L, lane);
                if (escL && r < kbad) ll += L.vval[r];
                if (escM && r2 < kbad) ml += L.vval[r2];
            }
        }
        ml += real ? 5u : 0u;
#ifdef EXP_PHASES
        PHC(11);
        if (parsed && !fast_vi) PH(8); else PH(0);
#endif

        // cursors: inclusive scans of (ll+ml) and ll
        uint32_t len = ll + ml;
        const uint32_t Eincl0 = wave_scan_add(len, lane);
        const uint32_t Lincl0 = wave_scan_add(ll, lane);
        const uint32_t est = p + (Eincl0 - len);  // where this sequence's literals land
        const uint32_t lst = lp + (Lincl0 - ll);  // its first literal
        // bounds (reference: OVERFLOW src/lib/zxc_decompress.c:1184, BAD_OFFSET :1190), all lanes, no branches
        const bool pseudo = valid & !real;  // whatever literals are left
        const bool lit_over = lst > n_lit;
        const uint32_t lit_left = n_lit - lst;
        bool ovf_real = (est > cap) | (len > cap - est) | lit_over | (ll > lit_left);
        bool cut4 = false;
        if (!strict) {
            const uint32_t gi = s & 3u;
            const int gl = lane - (int)gi;  // my group's first sequence (< 0: in an earlier batch)
            const uint32_t g_est = __shfl(est, gl < 0 ? 0 : gl), g_lst = __shfl(lst, gl < 0 ? 0 : gl);
            bool g4 = gl >= 0 ? (g_est + RD < cap && n_lit > RLM && g_lst < n_lit - RLM) : carry4x;
            g4 = g4 && real && S.n_seq - (s - gi) >= 4u;
            const uint32_t rawl = GHI ? raw_t >> 24 : raw_t >> 4;
            const uint32_t r1 = __shfl(rawl, (lane + 1) & 63), r2 = __shfl(rawl, (lane + 2) & 63), r3 = __shfl(rawl, (lane + 3) & 63);
            const uint32_t rest = (gi < 3u ? r1 : 0u) + (gi < 2u ? r2 : 0u) + (gi < 1u ? r3 : 0u);
            const bool r_lit = escL && !lit_over && ll + rest > lit_left;
            const bool r_dst = (escL | escM) && est <= cap && len + (3u - gi) * RW + 32u > cap - est;
            ovf_real |= g4 & (r_lit | r_dst);
            cut4 = seq_base + 64u < S.n_seq && lane + (int)(3u - gi) > 63;  // judged by the batch that holds the whole group
        }
        const bool ovf_pseudo = (est > cap) | lit_over | (lit_left > cap - est);
        const bool bad_off = off > est + ll + (DICT ? S.dict_size : 0u);
        int err = 0;
        err = (real & bad_off) ? E_BAD_OFFSET : err;
        err = ((real & ovf_real) | (pseudo & ovf_pseudo)) ? E_OVERFLOW : err;
        if (pseudo & !ovf_pseudo) { ll = lit_left; len = lit_left; }
        const uint32_t M = est + ll;
        const uint32_t E = est + len;

        // how many leading sequences fit a tile of TILE_MAX bytes (errors first, in order)
        if (cut4) err = 0;
        const uint64_t fits = __ballot(valid && !cut4 && err == 0 && (E - p) <= TILE_MAX);
        const uint64_t em = __ballot(err != 0);
        uint32_t k = (uint32_t)__ffsll((unsigned long long)~fits);  // 1-based index of first non-fitting lane
        k = k ? k - 1u : 64u;
        if (em) {
            const uint32_t e = (uint32_t)__ffsll((unsigned long long)em) - 1u;
            if (e <= k) return __shfl(err, (int)e);  // first failing sequence in stream order
        }

        PH(9);
        // the next batch starts at sequence seq_base + max(k, 1): request its tokens and offsets now
        uint32_t nraw_t, nraw_o;
        load_seq_raw<GHI>(S, seq_base + (k ? k : 1u) + (uint32_t)lane, nraw_t, nraw_o);
        // extras cursor after the k sequences consumed (the rest is re-parsed next turn), and its window
        if (parsed) {
            const uint32_t kk = k ? k : 1u;
            const uint64_t km = (kk >= 64u) ? ~0ull : ((1ull << kk) - 1ull);
            const uint32_t used = __popcll(mL & km) + __popcll(mM & km);
            if (fast_vi) cur += used;
            else if (kbad < used) { dead = 1; cur = S.ext_size; }
            else cur = uni(L.vpos[used]);
        }
        const int32_t xi0 = (int32_t)(cur + (uint32_t)lane), xi1 = xi0 + 64;
        const uint32_t nxw0 = ld8(S.ext + (xi0 < ext_last ? xi0 : ext_last));
        const uint32_t nxw1 = ld8(S.ext + (xi1 < ext_last ? xi1 : ext_last));
        PH(10);
        if (k == 0u) {
            // ---- one giant sequence (> TILE_MAX bytes): the whole wave walks it in pieces (plain stores)
            __builtin_amdgcn_s_waitcnt(0);  // (nothing stays on the compiler's scoreboard across the loop edge: see the flush below)
            const uint32_t gll = uni(ll), gml = uni(ml), goff = uni(off);
            uint32_t donel = 0;
            while (donel < gll) {
                const uint32_t n = (gll - donel < TILE_MAX) ? gll - donel : TILE_MAX;
                coop_copy<DICT>(L, O, p + donel, 0, S.lit + lp + donel, n, 0, lane);
                donel += n;
                flush_to(L, O, p + donel, lane);
            }
            if (gml) {
                __builtin_amdgcn_s_waitcnt(0);  // earlier flush stores must have left before reading them back
                coop_match<DICT>(L, O, p + gll, gml, goff, 0, true, lane);
            }
            p += gll + gml;
            lp += gll;
            k = 1;
            // restore the ring invariant: zero from p to the end of its 16-byte chunk; beyond that
            // the next batch zeroes what it needs
            z_end = (p + 15u) & ~15u;
            if (p + (uint32_t)lane < z_end) ring_wr8(L, p + (uint32_t)lane, 0u);
            PH(7);
        } else {
            const bool mine = (uint32_t)lane < k;
            const uint32_t tile_end = (uint32_t)__builtin_amdgcn_readlane((int)E, (int)(k - 1u));
            const uint32_t z_new = (tile_end + 15u) & ~15u;
#ifdef EXP_NO_FAR  // experiment only (wrong output): every source is taken from the ring — the launch time with no read-back traffic at all
            const uint32_t ring_lo = 0u;
#else
            const uint32_t ring_lo = z_new > RING_BYTES ? z_new - RING_BYTES : 0u;
#endif
            if (z_new > z_end) {
                ring_zero(L, z_end, z_new, lane);
                z_end = z_new;
            }

            // ---- literals, part 1. A run lands on its destination's dword grid: group g of the run is the
            // 16 bytes at output position (est & ~3) + 16 g, and they come from literal offset
            // lst - (est & 3) + 16 g — memory loads take any alignment, so no shift is needed at all.
            // The first three groups are requested now; the loads fly while the dependency analysis
            // below (registers and cross-lane traffic only) runs.
            const uint32_t la = est & 3u;
            const uint32_t le = la + ll;              // end of the run on its grid (bytes from the grid origin)
            const bool lshort = mine && ll != 0u && ll <= LIT_MED;
            const uint8_t* lsrc = S.lit + (int32_t)(lst - la);
            v4u lv0 = {0, 0, 0, 0}, lv1 = {0, 0, 0, 0};
#if LIT_PRELOAD >= 3
            v4u lv2 = {0, 0, 0, 0};
#endif
            if (lshort) {
                lv0 = ld128(lsrc);
                if (le > 16u) lv1 = ld128(lsrc + 16u);
#if LIT_PRELOAD >= 3
                if (le > 32u) lv2 = ld128(lsrc + 32u);
#endif
            }

            // ---- matches. Sequence i may only copy once every earlier match of this batch
            // that overlaps its source [qa, qb) is finished: those are lanes ja..jb.
            const uint32_t Es = mine ? E : 0xFFFFFFFFu, Ms = mine ? M : 0xFFFFFFFFu;
            const bool fromdict = DICT && off > M;  // source starts inside the dictionary prefix
            const uint32_t qa = M - off;    // (wraps negative then; only src_rd8 / coop paths read it)
            const uint32_t qb = fromdict ? ((off - M < ml) ? ((ml - (off - M) < M) ? ml - (off - M) : M) : 0u)
                                         : ((qa + ml < M) ? qa + ml : M);
            bool pending = mine && ml != 0u;
            uint64_t need = 0;
            uint32_t qsrc = qa;  // where the copy reads from (qa unless redirected)
            // A match lands on ITS destination's dword grid like a literal run: group g = 16 bytes at
            // (M & ~3) + 16 g, from source position qsrc - (M & 3) + 16 g.
            const uint32_t ma = M & 3u;
            const uint32_t me = ma + ml;
            const uint32_t mg = M - ma;          // grid origin (4-aligned output position)
            const bool overlap = off < ml;
#if FAR_EARLY == 2
            // Sources older than the ring come back from the block's own output through L2 (~2 K clocks under load).
            // Such a source ends at least 350 bytes before the batch (ring_lo <= p - 497, ml <= MATCH_MED): the lane
            // depends on nothing in this batch and is not redirected, and its bytes were flushed by an earlier batch
            // (one wave's stores and loads reach L2 in program order). Its first two groups are requested HERE, behind
            // the literal loads, so the round trip runs under the whole dependency analysis below.
            const bool pf = pending && !fromdict && ml <= MATCH_MED && (!overlap || off >= 16u) && qa < ring_lo && qa >= 4u &&
                            (qa - ma) + 32u <= O.out_pad;
            v4u fr0 = {0, 0, 0, 0}, fr1 = {0, 0, 0, 0};
            if (pf) {
                fr0 = __builtin_nontemporal_load((const v4u_unaligned*)(O.dst + (qa - ma)));
                if (me > 16u) fr1 = __builtin_nontemporal_load((const v4u_unaligned*)(O.dst + (qa - ma) + 16u));
            }
#endif
            {
                // all lanes run the same bpermute sequence; only lanes reaching into the batch use it
                const uint32_t ja = lanes_le(Es, fromdict ? 0u : qa);     // first lane with E > qa
                const uint32_t jbp = lanes_le(Ms, qb ? qb - 1u : 0u);    // number of lanes with M < qb
                if (pending && qb > p && jbp > ja)
                    need = ((jbp >= 64u) ? ~0ull : ((1ull << jbp) - 1ull)) & ~((1ull << ja) - 1ull);
                // Redirect: when my whole source sits inside ONE earlier match j of this batch (within
                // its first period) and j has no pending dependency itself, my bytes equal j's source
                // bytes at the same displacement: read those instead and drop the dependency. Each
                // pass collapses one more level of such containments.
                const int jl = (int)(ja & 63u);
                const uint32_t jM = __shfl(M, jl), jE = __shfl(E, jl), jo = __shfl(off, jl);
                const bool simple = need != 0ull && !fromdict && jbp == ja + 1u && off >= ml && qa >= jM && qb <= jE && qb - jM <= jo;
#pragma unroll
                for (int pass = 0; pass < REDIRECT_PASSES; pass++) {
                    const uint32_t jready = __shfl((uint32_t)(need == 0ull), jl);
                    const uint32_t jq = __shfl(qsrc, jl);
                    const uint32_t nq = jq + (qa - jM);
                    if (simple && need != 0ull && jready && (int32_t)nq >= 0) {  // not into the dictionary prefix
                        qsrc = nq;
                        need = 0;
                    }
                }
            }
            const uint32_t sg = qsrc - ma;       // the match's source on the destination's dword grid (may be "negative" by up to 3 for qsrc < 3: only masked-off bytes)
            // lane-per-sequence in 16-byte groups works whenever a group's source is complete before
            // the group is put: no overlap at all, or a period of at least one group (+ grid phase).
            const bool farsrc = qsrc < ring_lo;
            const bool stepable = !fromdict && ml <= MATCH_MED && (!overlap || off >= 16u) && !(farsrc && qsrc < 4u);
            const bool bytewise = !fromdict && overlap && off < 16u && ml <= BYTEWISE_MAX && !farsrc;
            const bool is_long = !stepable && !bytewise;
            bool far_waited = false;
            // Sources older than the ring come back from the block's own output through L2 (~700 clk). Such a
            // lane depends on nothing in this batch, so its first two groups are requested here: the literal
            // loads above are older (VMEM returns in order), so waiting for them below does not wait for
            // these, and the literal puts run under the round trip.
#if FAR_EARLY != 2
            const bool pf = pending && need == 0ull && stepable && farsrc && sg + 32u <= O.out_pad;
            v4u fr0 = {0, 0, 0, 0}, fr1 = {0, 0, 0, 0};
#endif
#if FAR_EARLY == 2
            __builtin_amdgcn_s_waitcnt(0);  // literal data and the far groups (also: every flush store has landed)
            far_waited = true;
#elif FAR_EARLY
            // Both groups are requested by EVERY lane (lanes without a far source read the block's first 32
            // bytes: one cache line, no divergence), right behind the literal loads and without waiting for
            // anything: a wave's own stores and loads reach L2 in program order, so the flush stores that wrote
            // those bytes are ahead of these loads, and with a fixed number of younger loads the compiler waits
            // for the literal data with vmcnt(2) — the far round trip runs under the literal puts.
            {
                const uint8_t* fa = O.dst + (pf ? sg : 0u);
                fr0 = __builtin_nontemporal_load((const v4u_unaligned*)fa);
                fr1 = __builtin_nontemporal_load((const v4u_unaligned*)(fa + 16u));
            }
#else
            // (unconditional wait: the literal data is needed next anyway, and with every older access known to
            // be complete on both paths the compiler does not force these loads to finish before the literal puts)
            PH(13);
            __builtin_amdgcn_s_waitcnt(0);  // also: the flush stores that wrote those bytes have landed
            far_waited = true;
            if (pf) {
                fr0 = __builtin_nontemporal_load((const v4u_unaligned*)(O.dst + sg));
#if FAR_PRELOAD >= 2
                if (me > 16u) fr1 = __builtin_nontemporal_load((const v4u_unaligned*)(O.dst + sg + 16u));
#endif
            }
#endif
            PH(2);
            // ---- literals, part 2: one masked ds_or group per 16 bytes of grid
            {
                const uint32_t lg = est - la;
                {
                    const uint32_t t = lshort ? (le < 16u ? le : 16u) : 0u;
                    v4u d = group_keep_first(L, lv0, t);
                    d.x &= head_mask(la);
                    ring_or_group(L, lg, d);
                }
                if (__ballot(lshort && le > 16u)) {
                    const uint32_t t = (lshort && le > 16u) ? (le - 16u < 16u ? le - 16u : 16u) : 0u;
                    ring_or_group(L, lg + 16u, group_keep_first(L, lv1, t));
                }
#if LIT_PRELOAD >= 3
                if (__ballot(lshort && le > 32u)) {
                    const uint32_t t = (lshort && le > 32u) ? (le - 32u < 16u ? le - 32u : 16u) : 0u;
                    ring_or_group(L, lg + 32u, group_keep_first(L, lv2, t));
                }
#endif
#if LIT_LOOP2
#pragma unroll 1
                for (uint32_t go = LIT_PRELOAD >= 3 ? 48u : 32u; go < LIT_MED + 4u; go += 32u) {  // two groups per step, both requested by every lane
                    const bool actA = lshort && le > go, actB = lshort && le > go + 16u;
                    if (__ballot(actA) == 0ull) break;
                    // (a load under a condition is waited for on the spot: idle lanes re-read the start of the literal stream)
                    const v4u lvA = ld128(actA ? lsrc + go : S.lit);
                    const v4u lvB = ld128(actB ? lsrc + go + 16u : S.lit);
                    const uint32_t tA = actA ? (le - go < 16u ? le - go : 16u) : 0u;
                    const uint32_t tB = actB ? (le - go - 16u < 16u ? le - go - 16u : 16u) : 0u;
                    ring_or_group(L, lg + go, group_keep_first(L, lvA, tA));
                    ring_or_group(L, lg + go + 16u, group_keep_first(L, lvB, tB));
                }
#else
#pragma unroll 1
                for (uint32_t go = 48u; go < LIT_MED + 4u; go += 16u) {
                    const bool act = lshort && le > go;
                    if (__ballot(act) == 0ull) break;
                    v4u lv = {0, 0, 0, 0};
                    if (act) lv = ld128(lsrc + go);
                    const uint32_t t = act ? (le - go < 16u ? le - go : 16u) : 0u;
                    ring_or_group(L, lg + go, group_keep_first(L, lv, t));
                }
#endif
                PH(7);  // (experiment builds: slot 7 = literal groups, slot 1 = long literals)
                uint64_t lm = __ballot(mine && ll > LIT_MED);
                while (lm) {
                    const int j = __ffsll((unsigned long long)lm) - 1;
                    lm &= lm - 1ull;
                    const uint32_t jl = __shfl(ll, j), je = __shfl(est, j), js = __shfl(lst, j);
                    coop_copy<DICT>(L, O, je, 0, S.lit + js, jl, 0, lane);
                }
                wave_lds_fence();
            }
            PH(1);
            for (uint32_t round = 0;; round++) {
                const uint64_t dm = __ballot(!pending);
                if (dm == ~0ull) break;
                if (round > 70u) return ZXC_DEV_E_INTERNAL;  // cannot happen: the lowest pending lane is always ready
                const bool can = pending && ((dm & need) == need);
                // lane-per-sequence copy, one 16-byte grid group per step (up to MATCH_MED bytes)
                const bool sa = can && stepable;
                if (__ballot(sa)) {
#pragma unroll 1
                    for (uint32_t go = 0; go < MATCH_MED + 4u; go += 16u) {
                        const bool act = sa && me > go;
                        if (__ballot(act) == 0ull) break;
                        const uint32_t q = sg + go;                       // source of this group
                        const bool usepf = act && pf && go < 16u * FAR_PRELOAD;  // requested before the literal puts
                        const bool isfar = act && !usepf && farsrc && q < ring_lo;  // (farsrc lanes have qsrc >= 4: q does not wrap)
                        v4u d = {0, 0, 0, 0};
                        if (usepf) d = go == 0u ? fr0 : fr1;
                        if (__ballot(isfar)) {
                            if (!far_waited) { __builtin_amdgcn_s_waitcnt(0); far_waited = true; }
                            if (isfar) d = far_rd128(O, q);
                        }
                        if (act && !isfar && !usepf) d = ring_rd128(L, q);
                        const uint32_t t = act ? (me - go < 16u ? me - go : 16u) : 0u;
                        d = group_keep_first(L, d, t);
                        if (go == 0u) d.x &= head_mask(ma);
                        ring_or_group(L, mg + go, d);
                    }
                }
                PH(3);
                // short period (off < 16, off < ml <= 32): byte loop over the period [M-off, M)
                const bool sb = can && bytewise;
                if (__ballot(sb)) {
                    uint32_t r = 0;
#pragma unroll 1
                    for (uint32_t t = 0; t < BYTEWISE_MAX; t++) {
                        const bool act = sb && t < ml;
                        if (__ballot(act) == 0ull) break;
                        if (act) {
                            ring_wr8(L, M + t, ring_rd8(L, qa + r));
                            r = (r + 1u == off) ? 0u : r + 1u;
                        }
                    }
                }
                PH(14);
                // long: the whole wave copies one match at a time
                uint64_t lm = __ballot(can && is_long);
                if (lm) wave_lds_fence();
                while (lm) {
                    const int j = __ffsll((unsigned long long)lm) - 1;
                    lm &= lm - 1ull;
                    const uint32_t jM = __shfl(M, j), jml = __shfl(ml, j), joff = __shfl(off, j);
                    if (jM - joff < ring_lo && !far_waited) { __builtin_amdgcn_s_waitcnt(0); far_waited = true; }
                    coop_match<DICT>(L, O, jM, jml, joff, ring_lo, false, lane);
                }
                if (can) pending = false;
                wave_lds_fence();
                PH(15);
                PHC(12);
                // Few sequences left (the typical batch has ~4 dependent ones spread over 1-2 more rounds):
                // finish them one by one in stream order with the whole wave, a byte per lane. In order,
                // every source is complete by construction, and one such copy costs a small fraction of
                // a full lane-per-sequence round.
                uint64_t pm = __ballot(pending);
                if (pm != 0ull && __popcll(pm) <= SPARSE_MAX) {
                    const uint64_t longm = __ballot(is_long);
                    while (pm) {
                        const int j = __ffsll((unsigned long long)pm) - 1;
                        pm &= pm - 1ull;
                        const uint32_t jM = (uint32_t)__builtin_amdgcn_readlane((int)M, j),
                                       jml = (uint32_t)__builtin_amdgcn_readlane((int)ml, j),
                                       joff = (uint32_t)__builtin_amdgcn_readlane((int)off, j);
                        const uint32_t jq = jM - joff;
                        if (((longm >> j) & 1ull) || jq < ring_lo) {
                            if (jq < ring_lo && !far_waited) { __builtin_amdgcn_s_waitcnt(0); far_waited = true; }
                            coop_match<DICT>(L, O, jM, jml, joff, ring_lo, false, lane);
                            continue;
                        }
                        const bool ovl = joff < jml;  // period joff: byte t comes from the first period, t mod joff
                        const float rcp = __builtin_amdgcn_rcpf((float)joff);
                        for (uint32_t t0 = 0; t0 < jml; t0 += 64u) {
                            const uint32_t t = t0 + (uint32_t)lane;
                            uint32_t r = t;
                            if (ovl) {
                                const uint32_t qd = (uint32_t)((float)t * rcp);
                                r = t - qd * joff;
                                if ((int32_t)r < 0) r += joff;
                                if (r >= joff) r -= joff;
                            }
                            if (t < jml) ring_wr8(L, jM + t, ring_rd8(L, jq + r));
                        }
                        wave_lds_fence();
                    }
                    pending = false;
#ifdef EXP_PHASES
                    PH(4);
#endif
                    break;
                }
            }
            p = tile_end;
            lp = (uint32_t)__builtin_amdgcn_readlane((int)(lst + ll), (int)(k - 1u));
            // Every load of this batch has been consumed by now, but some only under a condition, and the compiler
            // keeps those on its scoreboard: the first register reuse in the next batch would then wait for everything
            // older, i.e. for the flush stores below (~3 K clocks per batch, measured). Waiting here costs nothing
            // and leaves only the stores outstanding across the loop edge.
            __builtin_amdgcn_s_waitcnt(0);
            flush_to(L, O, p, lane);
            PH(5);
        }

#ifdef EXP_EXTRA_VALU  // experiment only: EXP_EXTRA_VALU independent VALU instructions per batch (is the kernel VALU-bound?)
        {
            uint32_t dummy = (uint32_t)lane;
#pragma unroll
            for (int i = 0; i < EXP_EXTRA_VALU; i++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(dummy));
            asm volatile("" ::"v"(dummy));
        }
#endif
#ifdef EXP_EXTRA_SLEEP  // experiment only: the wave sleeps 64 x EXP_EXTRA_SLEEP clocks per batch (is it latency-bound?)
        __builtin_amdgcn_s_sleep(EXP_EXTRA_SLEEP);
#endif
        if (!strict && ((seq_base + k) & 3u) != 0u) {
            const uint32_t g = (seq_base + k) & ~3u;  // the group the next batch starts inside
            if (g >= seq_base) {
                const uint32_t ge = (uint32_t)__builtin_amdgcn_readlane((int)est, (int)(g - seq_base)),
                               gl = (uint32_t)__builtin_amdgcn_readlane((int)lst, (int)(g - seq_base));
                carry4x = ge + RD < cap && n_lit > RLM && gl < n_lit - RLM;
            }
        }
        seq_base += k;
        raw_t = nraw_t;
        raw_o = nraw_o;
        xw0 = nxw0;
        xw1 = nxw1;
        wave_lds_fence();
        PH(6);
    }
#ifdef EXP_PHASES
    __builtin_amdgcn_s_waitcnt(0);
    if (lane < 16) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) v = (lane == i) ? ph[i] : v;
        ((uint32_t*)dst)[lane] = v;
    }
    return (int)p;
#endif
    // the last chunk may be partial: it only lives in the ring so far
    if ((p & 15u) != 0u && lane == 0) {
        const uint32_t cs = p & ~15u;
        for (uint32_t kk = 0; cs + kk < p && cs + kk < out_len; kk++) dst[cs + kk] = (uint8_t)ring_rd8(L, cs + kk);
    }
    return (int)p;
}

// The sequence executor of round 3 (8-waves-per-SIMD capable, far sources requested early, aligned loads, whole-KiB
// flushes). Both kernels run it: the lean kernel on raw sections, the full kernel on the sections it has expanded into its
// scratch slot; run_sequences() above remains for archives with a dictionary.
#ifndef LEAN_WAVES_PER_SIMD
#define LEAN_WAVES_PER_SIMD 6
#endif
#include "zxc_seq_lean.inc"

// Scratch (expanded literals / PivCo ping-pong / decoded tokens) is only needed by blocks with
// an RLE or PivCo section. Slots come from a small pool sized to the number of workgroups that
// can be resident at once, so a free one always exists: lane 0 claims one with atomicCAS.
struct ScratchPool {
    uint8_t* base;
    uint32_t stride;
    uint32_t* busy;
    uint32_t n_slots;
    int held;  // slot index or -1
};
__device__ __forceinline__ uint8_t* scratch_acquire(ScratchPool& sp, int lane) {
    if (sp.held < 0) {
        uint32_t got = 0;
        if (lane == 0) {
            uint32_t s = blockIdx.x % sp.n_slots;
            for (;;) {
                if (atomicCAS(sp.busy + s, 0u, 1u) == 0u) break;
                s = (s + 1u == sp.n_slots) ? 0u : s + 1u;
            }
            got = s;
        }
        sp.held = (int)uni(got);
    }
    return sp.base + (size_t)sp.held * sp.stride;
}
__device__ __forceinline__ void scratch_release(ScratchPool& sp, int lane) {
    if (sp.held >= 0) {
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) __hip_atomic_store(sp.busy + sp.held, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        sp.held = -1;
    }
}

// RLE literal section -> scratch (reference src/lib/zxc_decompress.c:906-975).
// Tokens form a chain (a raw token skips its payload), so finding them is sequential, but it is
// only "read a byte, add": the scalar unit walks the chain over a 256-byte window held one dword
// per lane (v_readlane), four windows prefetched ahead, and hands token k of a batch to lane k. The 64 tokens of a batch then expand in lockstep, every lane copying / filling
// its own 1..131 bytes with exact-length 16/8/4/2/1-byte global stores. Walking costs ~15 scalar
// instructions per token; the memory latency of the payload copies is paid once per 64 tokens.
__device__ __forceinline__ void st_bytes(uint8_t* d, const void* v, int nbytes) { __builtin_memcpy(d, v, nbytes); }

__device__ __forceinline__ uint32_t rle_window(const uint8_t* __restrict__ r, uint32_t rsize, uint32_t base, int lane) {
    const uint32_t a = base + 4u * (uint32_t)lane;
    if (a + 4u <= rsize) return ld32(r + a);
    uint32_t v = 0;
    for (uint32_t k = 0; k < 4u; k++)
        if (a + k < rsize) v |= ld8(r + a + k) << (8u * k);
    return v;
}

__device__ int rle_expand(const uint8_t* __restrict__ r_, uint32_t rsize, uint8_t* __restrict__ w_, uint32_t n, int lane) {
    const uint8_t* __restrict__ r = uni_ptr(r_);
    uint8_t* __restrict__ w = uni_ptr(w_);
    rsize = uni(rsize);
    n = uni(n);
    uint32_t pos = 0, dst = 0;  // wave-uniform: chain position in r, bytes produced
    uint32_t wbase = 0;
    uint32_t q0 = rle_window(r, rsize, 0u, lane), q1 = rle_window(r, rsize, 256u, lane),
             q2 = rle_window(r, rsize, 512u, lane), q3 = rle_window(r, rsize, 768u, lane);
    int rc = 0;
    while (pos < rsize && dst < n) {
        // ---- phase A: the next (up to) 64 tokens, token k -> lane k
        uint32_t tpos = 0, tdst = 0, tbyte = 0, ntok = 0;
        while (ntok < 64u && pos < rsize && dst < n) {
            while (pos - wbase >= 256u) {
                q0 = q1; q1 = q2; q2 = q3;
                wbase += 256u;
                q3 = rle_window(r, rsize, wbase + 768u, lane);
            }
            const uint32_t rel = pos - wbase;
            const uint32_t dw = (uint32_t)__builtin_amdgcn_readlane((int)q0, (int)(rel >> 2));
            const uint32_t tb = (dw >> (8u * (rel & 3u))) & 255u;
            const bool raw = !(tb & 0x80u);
            const bool me = (uint32_t)lane == ntok;  // one v_cmp + three v_cndmask with scalar sources
            tpos = me ? pos : tpos;
            tdst = me ? dst : tdst;
            tbyte = me ? tb : tbyte;
            dst += raw ? tb + 1u : (tb & 0x7Fu) + 4u;
            pos += raw ? tb + 2u : 2u;
            ntok++;
        }
        // ---- phase B: lane k expands token k
        const bool live = (uint32_t)lane < ntok;  // (the reference stops taking tokens once n bytes are out)
        const bool israw = !(tbyte & 0x80u);
        const uint32_t len = israw ? tbyte + 1u : (tbyte & 0x7Fu) + 4u;
        const uint32_t ri = tpos + 1u;
        const bool bad = live && ((n - tdst < len) || (israw ? (ri > rsize || rsize - ri < len) : (ri >= rsize)));
        if (__ballot(bad)) { rc = E_CORRUPT; break; }
        const uint8_t* sp = r + ri;
        uint8_t* dp = w + tdst;
        v4u fill = {0, 0, 0, 0};
        if (!israw) {
            const uint32_t b4 = (live ? ld8(sp) : 0u) * 0x01010101u;
            fill.x = b4; fill.y = b4; fill.z = b4; fill.w = b4;
        }
#pragma unroll 1
        for (uint32_t o = 0; o < 128u; o += 32u) {  // two 16-byte pieces in flight per step
            const bool a0 = live && o + 16u <= len, a1 = live && o + 32u <= len;
            if (__ballot(a0) == 0ull) break;
            v4u v0 = fill, v1 = fill;
            if (a0 && israw) v0 = ld128(sp + o);
            if (a1 && israw) v1 = ld128(sp + o + 16u);
            if (a0) st_bytes(dp + o, &v0, 16);
            if (a1) st_bytes(dp + o + 16u, &v1, 16);
        }
        if (live) {  // exact tail: 8 / 4 / 2 / 1 (and the 16-byte piece at 128 of a 131-byte run)
            uint32_t o = len & ~15u;
            if (len & 8u) { uint64_t v = israw ? ld64(sp + o) : ((uint64_t)fill.x << 32 | fill.x); st_bytes(dp + o, &v, 8); o += 8u; }
            if (len & 4u) { uint32_t v = israw ? ld32(sp + o) : fill.x; st_bytes(dp + o, &v, 4); o += 4u; }
            if (len & 2u) { uint16_t v = (uint16_t)(israw ? ld16(sp + o) : fill.x); st_bytes(dp + o, &v, 2); o += 2u; }
            if (len & 1u) { dp[o] = (uint8_t)(israw ? ld8(sp + o) : fill.x); }
        }
    }
    if (rc == 0 && dst != n) rc = E_CORRUPT;
    // every lane is about to read the scratch with ordinary (L1-cached) loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return rc;
}

template <bool DICT>
__device__ __forceinline__ int decode_lz_block(const uint8_t* data, uint32_t comp_sz, bool ghi, uint8_t* dst, uint32_t out_len,
                               uint32_t cap, uint32_t block_size, ScratchPool& pool, WaveLds& L, int lane,
                               uint32_t dbg, const uint8_t* dict, uint32_t dict_size, const uint8_t* dict_huf, const bool strict) {
    if (comp_sz < 12u) return E_BAD_HEADER;
    LzStreams S;
    S.dict = dict;
    S.dict_size = dict_size;
    S.n_seq = uni(ld32(data));
    S.n_lit = uni(ld32(data + 4));
    const uint32_t enc_lit = uni(ld8(data + 8)), enc_tok = uni(ld8(data + 9)), enc_off = uni(ld8(data + 11));
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
        if (!DICT) return run_sequences_lean<true>(S, dst, out_len, cap, reinterpret_cast<LeanLds&>(L), lane, strict);
        return run_sequences<DICT, true>(S, dst, out_len, cap, L, lane, strict);
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
#ifdef ZXC_LEAN_KERNEL  // the variant for blocks whose sections are all raw: no RLE / PivCo callees, no scratch
    if (enc_lit != 0u || enc_tok != 0u) return ZXC_DEV_E_UNSUPPORTED;
#else
    if (enc_lit == 2u || enc_lit == 3u) {
        if (lit_comp > avail) return E_CORRUPT;
        if (S.n_lit != 0u) {
            if (S.n_lit > cap) return E_DST_TOO_SMALL;
            if (enc_lit == 3u && !dict_huf) return E_DICT_REQUIRED;  // shared table comes with the dictionary
            if (S.n_lit > block_size) return E_CORRUPT;
            uint8_t* scratch = scratch_acquire(pool, lane) + 16;  // (the executor reads up to 3 bytes below a literal run)
            const int rc = pivco_decode(pdata, lit_comp, scratch, S.n_lit, scratch + ZXC_DEV_SLOT_REGION(block_size) - 16u,
                                  reinterpret_cast<PivLds&>(L), lane, enc_lit == 3u ? dict_huf : nullptr, dbg);
            if (rc != 0) return rc;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // literals are read back with L1-cached loads
            S.lit = scratch;
        }
    } else if (enc_lit == 1u) {
        if (S.n_lit != 0u) {
            if (S.n_lit > cap) return E_DST_TOO_SMALL;
            if (S.n_lit > block_size || lit_comp > avail) return E_CORRUPT;
            uint8_t* scratch = scratch_acquire(pool, lane) + 16;
            const int rc = rle_expand(pdata, lit_comp, scratch, S.n_lit, lane);
            if (rc != 0) return rc;
            S.lit = scratch;
        }
    } else if (enc_lit != 0u) {
        return E_CORRUPT;
    }
#endif
    const uint64_t sz_off = (uint64_t)S.n_seq * (enc_off ? 1u : 2u);
    const uint64_t consumed = (uint64_t)lit_comp + tok_comp + sz_off;
    if (consumed > avail || avail - lit_comp < 32u) return E_CORRUPT;
    S.tok = pdata + lit_comp;
#ifndef ZXC_LEAN_KERNEL
    if (enc_tok == 2u) {  // level 7: the token bytes are a PivCo section too
        if (S.n_seq > block_size / 5u + 16u) return E_CORRUPT;
        uint8_t* scratch = scratch_acquire(pool, lane);
        uint8_t* tokbuf = scratch + 2u * ZXC_DEV_SLOT_REGION(block_size);
        const int rc = pivco_decode(S.tok, tok_comp, tokbuf, S.n_seq, scratch + ZXC_DEV_SLOT_REGION(block_size),
                                    reinterpret_cast<PivLds&>(L), lane, nullptr, dbg);
        if (rc != 0) return rc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        S.tok = tokbuf;
    } else if (enc_tok != 0u) {
        return E_CORRUPT;
    }
#endif
    S.offs = pdata + lit_comp + tok_comp;
    S.off8 = enc_off;
    S.ext = S.offs + sz_off;
    S.ext_size = avail - (uint32_t)consumed;
#ifdef ZXC_EXPERIMENT
    if (dbg & DBG_NO_SEQ) return (int)out_len;
#endif
    if (!DICT) return run_sequences_lean<false>(S, dst, out_len, cap, reinterpret_cast<LeanLds&>(L), lane, strict);
    return run_sequences<DICT, false>(S, dst, out_len, cap, L, lane, strict);
}

#ifndef WAVES_PER_SIMD
#define WAVES_PER_SIMD 5   // 20 workgroups per CU fit the LDS (7.5 KiB each): keep everyone, callees included, within 96 VGPRs
#endif
template <bool DICT>
__device__ __forceinline__ void decode_one_block(const uint8_t* __restrict__ comp, const zxc_dev_job_t* __restrict__ jobs,
                                                 uint32_t n_jobs, uint8_t* __restrict__ out, int32_t* __restrict__ status,
                                                 uint32_t block_size, uint32_t trailer_bytes, uint8_t* __restrict__ scratch,
                                                 uint32_t scratch_stride, uint32_t dbg, uint32_t* __restrict__ slot_busy,
                                                 uint32_t n_slots, const uint32_t* __restrict__ order, uint32_t cap_override,
                                                 const uint8_t* __restrict__ dict, uint32_t dict_size,
                                                 const uint8_t* __restrict__ dict_huf, uint32_t slot) {
    // One workgroup (= one wavefront) per block: the hardware dispatcher hands out blocks as
    // wave slots free up, which is all the dynamic scheduling RAW-vs-dense blocks need.
#if defined(EXP_NO_PIV_LDS) || defined(ZXC_LEAN_KERNEL)  // (experiment: occupancy without the PivCo tables; lean variant: never decodes a section)
#ifdef LEAN_OWNER
    __shared__ union { WaveLds w; LeanLds l; } lds;
#else
    __shared__ union { WaveLds w; } lds;
#endif
#else
#ifdef LEAN_OWNER
    __shared__ union { WaveLds w; PivLds p; LeanLds l; } lds;  // (the owner executor's queue and window make its LDS the larger one)
#else
    __shared__ union { WaveLds w; PivLds p; } lds;  // the PivCo tables reuse the ring's LDS (never live together)
#endif
#endif
    WaveLds& L = lds.w;
    const int lane = threadIdx.x;
    if (slot >= n_jobs) return;
#ifdef EXP_PIV_PROF
    if (lane < 16) g_piv_prof[lane] = lane == 15 ? (uint32_t)__builtin_readcyclecounter() : 0u;
#endif
    // heaviest blocks first when the launch is long enough for the tail to matter (see zxc_order_* below)
    const uint32_t b = order ? uni(order[slot]) : slot;
#ifdef EXP_TIMES  // experiment only: status = start (hi 16) and duration (lo 16) in units of 32 ticks of the 100 MHz clock
    const uint64_t t_start = wall_clock64();
#endif
    // the reference decodes frames and seekable ranges with block_size + ZXC_DECOMPRESS_TAIL_PAD; only the strict
    // Block API (zxc_decompress_block_safe, src/lib/zxc_dispatch.c:1815-1858) passes the caller's exact capacity
    const uint32_t cap = cap_override ? cap_override : block_size + 2112u;
    const bool ck_here = !(trailer_bytes & ZXC_DEV_TRAILER_ELSEWHERE);  // (else zxc_block_checksum_kernel verifies, zxc_checksum_merge_kernel writes the verdicts)
    trailer_bytes &= ~ZXC_DEV_TRAILER_ELSEWHERE;
    ScratchPool pool = {scratch, scratch_stride, slot_busy, n_slots, -1};
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
        } else if (trailer_bytes && ck_here && wave_checksum32(src + 8, comp_sz, lane) != uni(ld32(src + 8 + comp_sz))) {
            rc = E_BAD_CHECKSUM;  // per-block checksum of the compressed payload (zxc_decompress.c:1662-1666)
        } else if (type == 1u || type == 2u) {
            rc = decode_lz_block<DICT>(src + 8, comp_sz, type == 2u, dst, out_len, cap, block_size, pool, L, lane, dbg, dict,
                                 dict_size, dict_huf, cap_override != 0u);
            scratch_release(pool, lane);
        } else if (type == 0u) {  // RAW: stored bytes
            if (comp_sz > cap) rc = E_DST_TOO_SMALL;
            else {
                const uint32_t n = comp_sz < out_len ? comp_sz : out_len;
                const uint8_t* s8 = src + 8;
                uint32_t i = 16u * lane;
                for (; i + 3072u + 16u <= n; i += 4096u) {  // 4 x 1 KiB in flight per wave
                    const v4u a0 = ld128(s8 + i), a1 = ld128(s8 + i + 1024u), a2 = ld128(s8 + i + 2048u),
                              a3 = ld128(s8 + i + 3072u);
                    *(v4u*)(dst + i) = a0;
                    *(v4u*)(dst + i + 1024u) = a1;
                    *(v4u*)(dst + i + 2048u) = a2;
                    *(v4u*)(dst + i + 3072u) = a3;
                }
                for (; i + 16u <= n; i += 1024u) *(v4u*)(dst + i) = ld128(s8 + i);
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
#ifdef EXP_TIMES
    __builtin_amdgcn_s_waitcnt(0);
    const uint64_t t_end = wall_clock64();
    rc = (int)((((uint32_t)(t_start >> 5) & 0xFFFFu) << 16) | (uint32_t)(((t_end - t_start) >> 5) & 0xFFFFu));
#endif
#ifdef EXP_PIV_PROF
    __builtin_amdgcn_s_waitcnt(0);
    if (lane < 16 && out_len >= 128u) ((uint32_t*)(dst + 64))[lane] = g_piv_prof[lane];
#endif
    if (lane == 0) status[b] = rc;
}

// Entry points. Archives without a dictionary (the common case and the benchmarked path) run the variant with every
// dictionary branch compiled out. `list` (or nullptr): big launches run the LEAN kernel below over every block first; it
// decodes the blocks whose sections are all raw and appends the others' positions to list[1 ..] (list[0] = their number):
// this kernel then runs with a fixed grid and walks that list.
extern "C" __global__ void __launch_bounds__(64, WAVES_PER_SIMD)
zxc_decode_blocks_kernel(const uint8_t* __restrict__ comp, const zxc_dev_job_t* __restrict__ jobs, uint32_t n_jobs,
                         uint8_t* __restrict__ out, int32_t* __restrict__ status, uint32_t block_size,
                         uint32_t trailer_bytes, uint8_t* __restrict__ scratch, uint32_t scratch_stride, uint32_t dbg,
                         uint32_t* __restrict__ slot_busy, uint32_t n_slots, const uint32_t* __restrict__ order,
                         uint32_t cap_override, uint32_t* __restrict__ list) {
    // (one call site for both modes. Plain launch: one block per workgroup, grid = n_jobs, the hardware dispatcher is the
    // dynamic scheduler. List mode: a fixed grid of workgroups PULLS list entries through a counter — list[0] = number of
    // entries, list[1] = next entry to hand out, entries from list[2] — so a slow block delays only its own workgroup.)
    if (!list) {
        decode_one_block<false>(comp, jobs, n_jobs, out, status, block_size, trailer_bytes, scratch, scratch_stride, dbg,
                                slot_busy, n_slots, order, cap_override, nullptr, 0u, nullptr, blockIdx.x);
        return;
    }
    const uint32_t n = uni(__hip_atomic_load(list, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (;;) {
        uint32_t i = 0;
        if (threadIdx.x == 0) i = atomicAdd(list + 1, 1u);
        i = uni(i);
        if (i >= n) break;
        decode_one_block<false>(comp, jobs, n_jobs, out, status, block_size, trailer_bytes, scratch, scratch_stride, dbg,
                                slot_busy, n_slots, order, cap_override, nullptr, 0u, nullptr, uni(list[2u + i]));
        wave_lds_fence();
    }
}

extern "C" __global__ void __launch_bounds__(64, WAVES_PER_SIMD)
zxc_decode_blocks_dict_kernel(const uint8_t* __restrict__ comp, const zxc_dev_job_t* __restrict__ jobs, uint32_t n_jobs,
                              uint8_t* __restrict__ out, int32_t* __restrict__ status, uint32_t block_size,
                              uint32_t trailer_bytes, uint8_t* __restrict__ scratch, uint32_t scratch_stride, uint32_t dbg,
                              uint32_t* __restrict__ slot_busy, uint32_t n_slots, const uint32_t* __restrict__ order,
                              uint32_t cap_override, const uint8_t* __restrict__ dict, uint32_t dict_size,
                              const uint8_t* __restrict__ dict_huf) {
    decode_one_block<true>(comp, jobs, n_jobs, out, status, block_size, trailer_bytes, scratch, scratch_stride, dbg,
                           slot_busy, n_slots, order, cap_override, dict, dict_size, dict_huf, blockIdx.x);
}

// Which kernel decodes a block: the lean one unless it is a GLO block with a coded literal / token section (every other
// block, malformed ones included, gets its verdict from the lean kernel). ONE predicate for the lean kernel and for the
// pass that builds the full kernel's list: a block must be taken by exactly one of them.
__device__ __forceinline__ bool block_needs_full_kernel(const uint8_t* __restrict__ src, uint32_t src_sz, uint32_t trailer_bytes) {
    if (src_sz < 8u + 12u) return false;
    const uint32_t comp_sz = ld32(src + 3);
    if (ld8(src) != 1u || comp_sz < 12u || (uint64_t)8u + comp_sz + trailer_bytes > src_sz) return false;
    return ld8(src + 16) != 0u || ld8(src + 17) != 0u;
}

// A GLO block with RLE-coded literals and raw tokens whose header holds up (the checks of decode_lz_block, in its order: none
// of them can fail for such a block): the lean kernel expands the literals itself into a slot of the scratch pool and runs
// the block like one with raw sections — 1.2 % of the level-3 blocks of the bench corpus, which cost 4 % of the launch while
// the one-wave full kernel ran beside the lean kernel for them (VERDICT r3 weak #2c). data = the block's payload (comp_sz >= 12).
__device__ __forceinline__ bool rle_block_for_lean(const uint8_t* __restrict__ data, uint32_t comp_sz, uint32_t block_size, uint32_t cap) {
    if (ld8(data + 8) != 1u || ld8(data + 9) != 0u || ld8(data + 11) > 1u || comp_sz < 16u) return false;
    const uint32_t n_seq = ld32(data), n_lit = ld32(data + 4), lit_comp = ld32(data + 12), avail = comp_sz - 16u;
    if (n_lit > cap || n_lit > block_size || lit_comp > avail) return false;
    const uint64_t consumed = (uint64_t)lit_comp + n_seq + (uint64_t)n_seq * (ld8(data + 11) ? 1u : 2u);
    return consumed <= avail && avail - lit_comp >= 32u;
}

// The class of a block in a launch without a dictionary (zxc_dev.h). PRE = a GLO block whose coded sections are PivCo
// (no RLE) and fit a workgroup decoder's LDS, with every header field the section kernels and the lean kernel rely on
// already valid; anything else that needs the full kernel — malformed headers included, it names their errors — is FULL.
struct BlockClass {
    uint32_t cls;
    uint32_t lit16, tok16;      // scratch for the decoded sections, 16-byte units
    uint32_t lit_cls, tok_cls;  // size class of the section's work list (pdir_class), 3 = not coded
    uint32_t lit_at, lit_psize, n_lit, tok_at, tok_psize, n_seq;  // payload offsets from the block's first byte, sizes, symbols
};
__device__ __forceinline__ BlockClass classify_block(const uint8_t* __restrict__ src, uint32_t src_sz, uint32_t trailer_bytes,
                                                     uint32_t block_size, uint32_t cap) {
    BlockClass r = {ZXC_DEV_CLS_LEAN, 0, 0, 3, 3, 0, 0, 0, 0, 0, 0};
    if (!block_needs_full_kernel(src, src_sz, trailer_bytes)) return r;
#ifndef EXP_RLE_FULL  // (A/B: RLE blocks to the full kernel, as in round 3)
    if (rle_block_for_lean(src + 8, ld32(src + 3), block_size, cap)) {
        r.cls = ZXC_DEV_CLS_LEAN_RLE;
        r.n_lit = ld32(src + 12);
        r.lit16 = r.n_lit ? (16u + r.n_lit + 64u + 15u) >> 4 : 0u;  // 16 bytes in front (the executor reads up to 3 bytes below a literal run), 64 behind
        return r;
    }
#endif
    r.cls = ZXC_DEV_CLS_FULL;
    const uint8_t* data = src + 8;
    const uint32_t comp_sz = ld32(src + 3);
    const uint32_t n_seq = ld32(data), n_lit = ld32(data + 4), enc_lit = ld8(data + 8), enc_tok = ld8(data + 9), enc_off = ld8(data + 11);
    if ((enc_lit != 0u && enc_lit != 2u) || (enc_tok != 0u && enc_tok != 2u) || enc_off > 1u) return r;
    const uint32_t desc = (enc_lit != 0u ? 4u : 0u) + (enc_tok == 2u ? 4u : 0u);
    if (comp_sz < 12u + desc) return r;
    uint32_t lit_comp = n_lit, tok_comp = n_seq;
    const uint8_t* d = data + 12;
    if (enc_lit != 0u) { lit_comp = ld32(d); d += 4; }
    if (enc_tok == 2u) { tok_comp = ld32(d); d += 4; }
    const uint32_t avail = comp_sz - 12u - desc;
    if (lit_comp > avail) return r;
    if (enc_lit == 2u && (n_lit == 0u || n_lit > cap || n_lit > block_size || n_lit > PDIR_N_MAX || lit_comp < 128u || pdir_class(lit_comp - 128u) > 2u))
        return r;
    const uint64_t consumed = (uint64_t)lit_comp + tok_comp + (uint64_t)n_seq * (enc_off ? 1u : 2u);
    if (consumed > avail || avail - lit_comp < 32u) return r;
    if (enc_tok == 2u && (n_seq == 0u || n_seq > block_size / 5u + 16u || n_seq > PDIR_N_MAX || tok_comp < 128u || pdir_class(tok_comp - 128u) > 2u))
        return r;
    r.cls = ZXC_DEV_CLS_PRE;
    if (enc_lit == 2u) {
        r.lit16 = (16u + n_lit + 64u + 15u) >> 4;  // 16 bytes in front (the executor reads up to 3 bytes below a literal run), 64 behind
        r.lit_cls = pdir_class(lit_comp - 128u);
        r.lit_at = 8u + 12u + desc;
        r.lit_psize = lit_comp;
        r.n_lit = n_lit;
    }
    if (enc_tok == 2u) {
        r.tok16 = (n_seq + 64u + 15u) >> 4;
        r.tok_cls = pdir_class(tok_comp - 128u);
        r.tok_at = 8u + 12u + desc + lit_comp;
        r.tok_psize = tok_comp;
        r.n_seq = n_seq;
    }
    return r;
}

// ------------------------------------------------------------------ the lean kernel
// One wavefront per block like the full kernel, built for LEAN_WAVES_PER_SIMD waves per SIMD (<= 64 VGPRs, < 5 KiB LDS):
// RAW blocks and GLO / GHI blocks with raw sections, no checksum, no dictionary. Two entries share this body. The first
// runs over every block of the launch and takes the LEAN class (zxc_dev.h); the second pulls the PRE blocks from their list
// once the section kernels have decoded their coded sections into pscratch (PRE = true).
template <bool PRE>
__device__ __forceinline__ void lean_one_block(const uint8_t* __restrict__ comp, const zxc_dev_job_t* __restrict__ jobs,
                                               uint8_t* __restrict__ out, int32_t* __restrict__ status, uint32_t block_size,
                                               uint32_t cap_override, uint32_t trailer_bytes, uint32_t b,
                                               const zxc_dev_pre_t* __restrict__ pre, const uint8_t* __restrict__ pscratch, LeanLds& L,
                                               int lane, const uint8_t* __restrict__ rle_slot, int rle_rc) {
    // (cap_override: only ever 0 here. A launch with the strict capacity of zxc_decompress_block_safe — exact checks, none of
    // the reference's 4x-batch reserve — goes to the full kernel alone: zxc_hip_shim.hip. The argument stays for the ABI.)
    const uint32_t cap = cap_override ? cap_override : block_size + 2112u;
    const bool ck_here = !(trailer_bytes & ZXC_DEV_TRAILER_ELSEWHERE);
    trailer_bytes &= ~ZXC_DEV_TRAILER_ELSEWHERE;
#ifdef EXP_TIMES  // experiment only (tools/blocktimes.py): status = start (hi 16) and duration (lo 16) in units of 32 ticks of the 100 MHz clock
    const uint64_t t_start = wall_clock64();
#endif
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
        } else if (trailer_bytes && ck_here && wave_checksum32(src + 8, comp_sz, lane) != uni(ld32(src + 8 + comp_sz))) {
            rc = E_BAD_CHECKSUM;  // per-block checksum of the compressed payload (zxc_decompress.c:1662-1666)
        } else if (PRE) {  // (a GLO block, by classify_block: nothing else is compiled into the second entry)
            rc = type == 1u ? decode_lz_block_lean(src + 8, comp_sz, false, dst, out_len, cap, L, lane, pre + b, pscratch, false) : ZXC_DEV_E_INTERNAL;
        } else if (type == 1u || type == 2u) {
            // rle_slot != nullptr: a LEAN_RLE block (header checked by classify_block = rle_block_for_lean); zxc_rle_expand_kernel
            // has expanded its literals to rle_slot + 16 (the executor reads up to 3 bytes below a literal run): rle_rc = its verdict
            rc = rle_slot ? rle_rc : 0;
            if (rc == 0) rc = decode_lz_block_lean(src + 8, comp_sz, type == 2u, dst, out_len, cap, L, lane, nullptr, pscratch, false,
                                                   rle_slot ? rle_slot + 16 : nullptr, rle_slot != nullptr);
        } else if (type == 0u) {  // RAW: stored bytes
            if (comp_sz > cap) rc = E_DST_TOO_SMALL;
            else {
                const uint32_t n = comp_sz < out_len ? comp_sz : out_len;
                const uint8_t* s8 = src + 8;
                uint32_t i = 16u * lane;
                for (; i + 3072u + 16u <= n; i += 4096u) {  // 4 x 1 KiB in flight per wave
                    const v4u a0 = ld128(s8 + i), a1 = ld128(s8 + i + 1024u), a2 = ld128(s8 + i + 2048u),
                              a3 = ld128(s8 + i + 3072u);
                    *(v4u*)(dst + i) = a0;
                    *(v4u*)(dst + i + 1024u) = a1;
                    *(v4u*)(dst + i + 2048u) = a2;
                    *(v4u*)(dst + i + 3072u) = a3;
                }
                for (; i + 16u <= n; i += 1024u) *(v4u*)(dst + i) = ld128(s8 + i);
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
#ifdef EXP_TIMES
    __builtin_amdgcn_s_waitcnt(0);
    {
        const uint64_t t_end = wall_clock64();
        rc = (int)((((uint32_t)(t_start >> 5) & 0xFFFFu) << 16) | (uint32_t)(((t_end - t_start) >> 5) & 0xFFFFu));
    }
#endif
    if (lane == 0) status[b] = rc == ZXC_DEV_DEFER ? ZXC_DEV_E_INTERNAL : rc;  // (DEFER cannot happen: see classify_block)
}

extern "C" __global__ void __launch_bounds__(64, LEAN_WAVES_PER_SIMD)
zxc_decode_blocks_lean_kernel(const uint8_t* __restrict__ comp, const zxc_dev_job_t* __restrict__ jobs, uint32_t n_jobs,
                              uint8_t* __restrict__ out, int32_t* __restrict__ status, uint32_t block_size,
                              const uint32_t* __restrict__ order, uint32_t cap_override, uint32_t trailer_bytes,
                              const zxc_dev_pre_t* __restrict__ pre, const uint8_t* __restrict__ rscratch) {
    __shared__ LeanLds L;
    if (blockIdx.x >= n_jobs) return;
    const uint32_t b = order ? uni(order[blockIdx.x]) : blockIdx.x;
    const uint32_t cls = uni(pre[b].cls);
    if (cls != ZXC_DEV_CLS_LEAN && cls != ZXC_DEV_CLS_LEAN_RLE) return;  // on the full kernel's list, or on the second entry's (zxc_order_scatter_kernel)
    // (rscratch: the launch's scratch for the expanded literals of LEAN_RLE blocks, handed out by the launch-order pass and
    // filled by zxc_rle_expand_kernel in front of this kernel)
    const uint8_t* rle_slot = cls == ZXC_DEV_CLS_LEAN_RLE ? rscratch + 16ull * uni(pre[b].lit_off) : nullptr;
    const int rle_rc = cls == ZXC_DEV_CLS_LEAN_RLE ? (int)(int16_t)uni((uint32_t)(uint16_t)pre[b].rc_lit) : 0;
    lean_one_block<false>(comp, jobs, out, status, block_size, cap_override, trailer_bytes, b, nullptr, nullptr, L, threadIdx.x, rle_slot, rle_rc);
}

// The literals of the launch's LEAN_RLE blocks, expanded into their shares of rscratch: the blocks are listed by
// zxc_order_scatter_kernel (hdr[0] = entries; job indices from entries_last backwards). A small fixed grid in front of the lean
// kernel on its stream, block i to workgroup i mod grid (no atomic hand-out: a wave-uniform pull loop around a one-lane atomic
// hung on the device here, profiles/r4b_rle_variants.log). What round 4 measured on the way (same log): the expansion inlined
// into the lean kernel costs every block of a launch 5 % (a third inlined copy of the executor); the blocks on a helper stream
// beside the lean kernel (own entry behind the expansion) start late and end the launch, like the one-wave full kernel did in
// round 3 (these blocks are the corpus' heaviest: ~5 500 sequences, ~400 literal bytes in ~40 RLE tokens).
extern "C" __global__ void __launch_bounds__(64)
zxc_rle_expand_kernel(const uint8_t* __restrict__ comp, const zxc_dev_job_t* __restrict__ jobs, zxc_dev_pre_t* __restrict__ pre,
                      uint8_t* __restrict__ rscratch, const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ entries_last) {
    const int lane = threadIdx.x;
    const uint32_t n = uni(hdr[0]);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {  // (a small fixed grid; block i to workgroup i mod grid)
        const uint32_t b = uni(*(entries_last - i));
        const uint8_t* src = comp + jobs[b].comp_off;
        const uint32_t n_lit = uni(ld32(src + 12));
        int rc = 0;
        if (n_lit != 0u) rc = rle_expand(src + 8 + 16, uni(ld32(src + 20)), rscratch + 16ull * uni(pre[b].lit_off) + 16u, n_lit, lane);
        if (lane == 0) pre[b].rc_lit = (int16_t)rc;
    }
}

// hdr[0] = PRE blocks listed, entries = their job indices; launched with one workgroup per block of the launch, the ones beyond
// the list leave at once (a fixed grid pulling entries in a loop kept 26 VGPRs of loop state in scratch memory: -20 %).
extern "C" __global__ void __launch_bounds__(64, LEAN_WAVES_PER_SIMD)
zxc_decode_blocks_lean_pre_kernel(const uint8_t* __restrict__ comp, const zxc_dev_job_t* __restrict__ jobs, uint8_t* __restrict__ out,
                                  int32_t* __restrict__ status, uint32_t block_size, uint32_t cap_override, uint32_t trailer_bytes,
                                  const zxc_dev_pre_t* __restrict__ pre, const uint8_t* __restrict__ pscratch,
                                  const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ entries) {
    __shared__ LeanLds L;
    if (blockIdx.x >= uni(hdr[0])) return;
    lean_one_block<true>(comp, jobs, out, status, block_size, cap_override, trailer_bytes, uni(entries[blockIdx.x]), pre, pscratch, L, threadIdx.x, nullptr, 0);
}

// ------------------------------------------------------------------ per-block checksums beside the decode (round 6)
// Nine blocks per wavefront (zxc_rapidhash.inc: group_checksum32), in launch order (neighbours are of similar weight). ck_bad[b] = 1 when
// block b carries a checksum that does not match its payload — and only then: a block whose header the decode kernels refuse before they
// would look at the checksum (too small, size beyond its bytes) keeps THEIR verdict (reference order: zxc_decompress.c:1655-1666).
extern "C" __global__ void __launch_bounds__(64)
zxc_block_checksum_kernel(const uint8_t* __restrict__ comp, const zxc_dev_job_t* __restrict__ jobs, uint32_t n_jobs,
                          const uint32_t* __restrict__ order, uint8_t* __restrict__ ck_bad) {
    const int lane = threadIdx.x;
    const uint32_t slot = blockIdx.x * 9u + (uint32_t)lane / 7u;
    const bool in = lane < 63 && slot < n_jobs;
    const uint32_t b = in ? (order ? order[slot] : slot) : 0u;
    const uint8_t* src = comp + jobs[b].comp_off;
    const uint32_t src_sz = in ? jobs[b].comp_size : 0u;
    uint32_t comp_sz = 0;
    bool on = false;
    if (src_sz >= 8u) {
        comp_sz = ld32(src + 3);
        on = (uint64_t)8u + comp_sz + 4u <= src_sz;
    }
    const uint32_t h = group_checksum32(src + 8, on ? comp_sz : 0u, lane, on);
    if (on && lane % 7 == 0) ck_bad[b] = h != ld32(src + 8 + comp_sz) ? 1u : 0u;
    else if (in && !on && lane % 7 == 0) ck_bad[b] = 0u;
}
extern "C" __global__ void __launch_bounds__(256)
zxc_checksum_merge_kernel(const uint8_t* __restrict__ ck_bad, int32_t* __restrict__ status, uint32_t n_jobs) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n_jobs && ck_bad[i]) status[i] = E_BAD_CHECKSUM;
}

// ------------------------------------------------------------------ launch order
// A launch ends when its slowest block ends, and block cost varies ~3x with the number of
// sequences and the literal coding. For launches of many rounds the blocks are dispatched
// heaviest-first (longest-processing-time order): a 64-bucket counting sort on a cost estimate
// read from each block header. order[0 .. n) = job indices; hist[0..64) counts, hist[64..128) cursors.
__device__ __forceinline__ uint32_t order_bucket(const uint8_t* __restrict__ comp, const zxc_dev_job_t& j, uint32_t block_size) {
    uint32_t cost = 0;
    if (j.comp_size >= 8u + 12u) {
        const uint8_t* h = comp + j.comp_off;
        const uint32_t type = ld8(h);
        if (type == 1u || type == 2u) {
            const uint32_t n_seq = ld32(h + 8), n_lit = ld32(h + 12), enc_lit = ld8(h + 16), enc_tok = ld8(h + 17);
            cost = n_seq + (enc_tok == 2u ? n_seq : 0u) + (enc_lit == 1u ? n_lit >> 4 : 0u) + (enc_lit >= 2u ? n_lit >> 2 : 0u);
        }
    }
    const uint64_t q = (uint64_t)cost * 320u / block_size;  // cost tops out near block_size / 5
    return 63u - (q > 63u ? 63u : (uint32_t)q);              // bucket 0 = heaviest
}

extern "C" __global__ void __launch_bounds__(256)
zxc_order_hist_kernel(const uint8_t* __restrict__ comp, const zxc_dev_job_t* __restrict__ jobs, uint32_t n_jobs,
                      uint32_t block_size, uint32_t* __restrict__ hist) {
    __shared__ uint32_t cnt[64];  // per-workgroup counts first: 64 global atomics per workgroup, not 256
    if (threadIdx.x < 64u) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n_jobs) atomicAdd(cnt + order_bucket(comp, jobs[i], block_size), 1u);
    __syncthreads();
    if (threadIdx.x < 64u && cnt[threadIdx.x]) atomicAdd(hist + threadIdx.x, cnt[threadIdx.x]);
}

// Work lists of a two-pass launch (list != nullptr), filled here from the block headers: the full kernel's (list[0] = entries,
// list[1] = 0, positions in launch order from list[2]); behind order[]: ctl (zxc_dev.h), the PRE blocks' job indices, one array
// of section records per size class (2 n each) and pre[]. Counters are bumped once per workgroup, not per block.
extern "C" __global__ void __launch_bounds__(256)
zxc_order_scatter_kernel(const uint8_t* __restrict__ comp, const zxc_dev_job_t* __restrict__ jobs, uint32_t n_jobs,
                         uint32_t block_size, uint32_t* __restrict__ hist, uint32_t* __restrict__ order,
                         uint32_t* __restrict__ list, uint32_t trailer_bytes, zxc_dev_pre_t* __restrict__ pre,
                         uint32_t* __restrict__ ctl, uint32_t* __restrict__ pre_entries, zxc_dev_sec_t* __restrict__ secs,
                         uint32_t pscratch_cap16, uint32_t cap, uint32_t rscratch_cap16) {
    __shared__ uint32_t cnt[64], base[64];
    __shared__ uint32_t wg_cnt[8], wg_base[8];  // 0: scratch units, 1: PRE blocks, 2: FULL blocks, 3..5: sections per size class, 6: RLE scratch units, 7: LEAN_RLE blocks
    __shared__ uint32_t wg_fit, wg_rle_fit;
    if (threadIdx.x < 64u) cnt[threadIdx.x] = 0;
    if (threadIdx.x < 8u) wg_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t bk = 0, rank = 0;
    if (i < n_jobs) {
        bk = order_bucket(comp, jobs[i], block_size);
        rank = atomicAdd(cnt + bk, 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64u) {  // this workgroup's slice of every bucket it touches
        uint32_t s = 0;
        for (uint32_t k = 0; k < threadIdx.x; k++) s += hist[k];
        base[threadIdx.x] = cnt[threadIdx.x] ? s + atomicAdd(hist + 64u + threadIdx.x, cnt[threadIdx.x]) : 0u;
    }
    __syncthreads();
    if (i < n_jobs) order[base[bk] + rank] = i;
    if (!list) return;
    BlockClass c = {ZXC_DEV_CLS_LEAN, 0, 0, 3, 3, 0, 0, 0, 0, 0, 0};
    uint32_t my_off = 0, my_pre = 0, my_full = 0, my_lit = 0, my_tok = 0;
    if (i < n_jobs) {
        c = classify_block(comp + jobs[i].comp_off, jobs[i].comp_size, trailer_bytes, block_size, cap);
        if (c.cls == ZXC_DEV_CLS_PRE) {
            my_off = atomicAdd(wg_cnt + 0, c.lit16 + c.tok16);
            my_pre = atomicAdd(wg_cnt + 1, 1u);
            if (c.lit_cls < 3u) my_lit = atomicAdd(wg_cnt + 3u + c.lit_cls, 1u);
            if (c.tok_cls < 3u) my_tok = atomicAdd(wg_cnt + 3u + c.tok_cls, 1u);
        } else if (c.cls == ZXC_DEV_CLS_FULL) {
            my_full = atomicAdd(wg_cnt + 2, 1u);
        } else if (c.cls == ZXC_DEV_CLS_LEAN_RLE) {
            my_off = atomicAdd(wg_cnt + 6, c.lit16);
            my_pre = atomicAdd(wg_cnt + 7, 1u);  // (its rank among this workgroup's LEAN_RLE blocks: the fallback's list position)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // RLE scratch for the LEAN_RLE blocks of this workgroup: the same saturating cursor as below; no room -> the full kernel
        // (no scratch at all — first launch on a stream, >= 16 384 jobs, allocation failure: even a LEAN_RLE block without literals
        //  must not stay here, the lean kernel tells such a block by its scratch pointer: reference accepts it, zxc_decompress.c:906-907)
        bool rfit = rscratch_cap16 != 0u;
        wg_base[6] = 0;
        if (wg_cnt[7]) {
            atomicAdd(ctl + ZXC_DEV_CTL_RLE_WANTED, wg_cnt[6] + 1u);  // (+ 1: a launch whose RLE blocks all have n_lit == 0 still reads as "some")
            if (wg_cnt[6]) {
                uint32_t old = __hip_atomic_load(ctl + ZXC_DEV_CTL_RLE_CURSOR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (;;) {
                    if (old > rscratch_cap16) { rfit = false; break; }
                    const uint32_t seen = atomicCAS(ctl + ZXC_DEV_CTL_RLE_CURSOR, old, old + wg_cnt[6]);
                    if (seen == old) { rfit = (uint64_t)old + wg_cnt[6] <= rscratch_cap16; break; }
                    old = seen;
                }
                wg_base[6] = old;
            }
            if (!rfit) wg_cnt[2] += wg_cnt[7];  // they join this workgroup's FULL blocks (behind them)
            else wg_base[7] = atomicAdd(ctl + ZXC_DEV_CTL_RLE_LIST, wg_cnt[7]);
        }
        wg_rle_fit = rfit ? 1u : 0u;
        bool fit = true;
        wg_base[0] = 0;
        if (wg_cnt[0]) {
            // The cursor only moves while it is inside the scratch, so it can never wrap (a launch of ~870 K level-7 blocks would
            // add up to more than 2^32 units otherwise): the first request that does not fit leaves it beyond the capacity, and
            // exhaustion is sticky from then on. (scratch exhausted: this workgroup's PRE blocks go to the full kernel and its slot pool)
            uint32_t old = __hip_atomic_load(ctl + ZXC_DEV_CTL_CURSOR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                if (old > pscratch_cap16) { fit = false; break; }
                const uint32_t seen = atomicCAS(ctl + ZXC_DEV_CTL_CURSOR, old, old + wg_cnt[0]);  // (old <= 2^26, a workgroup asks for < 2^27)
                if (seen == old) { fit = (uint64_t)old + wg_cnt[0] <= pscratch_cap16; break; }
                old = seen;
            }
            wg_base[0] = old;
        }
        wg_fit = fit ? 1u : 0u;
        if (wg_cnt[1]) atomicAdd(ctl + ZXC_DEV_CTL_WANTED, wg_cnt[1]);
        if (fit) {
            wg_base[1] = wg_cnt[1] ? atomicAdd(ctl + ZXC_DEV_CTL_PRE, wg_cnt[1]) : 0u;
            wg_base[2] = wg_cnt[2] ? atomicAdd(list, wg_cnt[2]) : 0u;
            for (uint32_t k = 0; k < 3u; k++) wg_base[3u + k] = wg_cnt[3u + k] ? atomicAdd(ctl + ZXC_DEV_CTL_SEC + 2u * k, wg_cnt[3u + k]) : 0u;
        } else {
            wg_base[2] = atomicAdd(list, wg_cnt[2] + wg_cnt[1]);
        }
    }
    __syncthreads();
    if (i >= n_jobs) return;
    uint32_t cls = c.cls;
    if (cls == ZXC_DEV_CLS_PRE && !wg_fit) {
        cls = ZXC_DEV_CLS_FULL;
        my_full = wg_cnt[2] + my_pre;
    }
    if (cls == ZXC_DEV_CLS_LEAN_RLE) {
        if (wg_rle_fit) {
            pre_entries[n_jobs - 1u - (wg_base[7] + my_pre)] = i;  // (the PRE blocks' entries grow from the front: the classes are disjoint)
            pre[i].lit_off = wg_base[6] + my_off;
            pre[i].tok_off = 0;
            pre[i].rc_lit = 0;
            pre[i].rc_tok = 0;
            pre[i].cls = cls;
            return;
        }
        cls = ZXC_DEV_CLS_FULL;
        my_full = wg_cnt[2] - wg_cnt[7] + my_pre;  // (wg_cnt[2] already includes this workgroup's displaced LEAN_RLE blocks)
    }
    const uint32_t off = wg_base[0] + my_off;
    pre[i].lit_off = off;
    pre[i].tok_off = off + c.lit16;
    pre[i].rc_lit = 0;
    pre[i].rc_tok = 0;
    pre[i].cls = cls;
    if (cls == ZXC_DEV_CLS_FULL) {
        list[2u + wg_base[2] + my_full] = base[bk] + rank;
    } else if (cls == ZXC_DEV_CLS_PRE) {
        pre_entries[wg_base[1] + my_pre] = i;
        if (c.lit_cls < 3u) {
            zxc_dev_sec_t& s = secs[(size_t)c.lit_cls * 2u * n_jobs + wg_base[3u + c.lit_cls] + my_lit];
            s.src_off = jobs[i].comp_off + c.lit_at;
            s.psize = c.lit_psize;
            s.n = c.n_lit;
            s.out_off4 = 4u * off + 4u;
            s.rc_slot = 8u * i + 4u;
        }
        if (c.tok_cls < 3u) {
            zxc_dev_sec_t& s = secs[(size_t)c.tok_cls * 2u * n_jobs + wg_base[3u + c.tok_cls] + my_tok];
            s.src_off = jobs[i].comp_off + c.tok_at;
            s.psize = c.tok_psize;
            s.n = c.n_seq;
            s.out_off4 = 4u * (off + c.lit16);
            s.rc_slot = 8u * i + 5u;
        }
    }
}
