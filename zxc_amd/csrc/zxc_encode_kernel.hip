// zxc_encode_kernel.hip — per-block LZ77 hash-chain match finder + GLO / GHI serialiser for gfx950.
//
// Reference being replaced: zxc_compress_chunk_wrapper (src/lib/zxc_compress.c:2041-2074) →
// zxc_encode_block_glo (:1124-1799) / zxc_encode_block_ghi (:1820-1987), whose hot loop is
// zxc_lz77_find_best_match (:185-547): head table + chain table walked for at most search_depth
// candidates, stop at sufficient_len, backward extension to the anchor, lazy probes at ip+1 / ip+2
// (per-level parameters src/lib/zxc_internal.h:965-979). The CPU loop is sequential (what gets inserted
// depends on what was matched); here the SAME search structure is filled and walked 64 positions at a time:
//
// One wavefront per block, 64 positions per step, one per lane:
//   1. hash (5 bytes, or 4 bytes at levels 1-2, the reference's two hash functions) -> head[] in LDS = the
//      most recent position with that hash; chain[] in LDS (a ring over the last 2^CWB positions) links every
//      position to the previous one with the same hash. EVERY position is inserted (the CPU inserts only the
//      positions its parse visits), so a chain here is denser than the reference's and needs fewer steps;
//   2. each lane walks its own chain for up to `depth` candidates, three to six per round: the links are LDS reads,
//      32 bytes of every candidate are requested as soon as its distance is known (one memory round trip per
//      round), compared with 64-bit XOR + ctz, the longest kept; candidates still equal after 32 bytes are
//      extended together, 16 bytes per step (A/B: 32 bytes per step with every load unconditional is 3 % slower
//      at level 3, the same at levels 5-7); the walk stops at `sufficient` bytes like the reference's;
//   3. the chunk's positions are published: chain link = distance to the old head, head = own position; lanes
//      that share a bucket resolve it deterministically (highest position wins, re-checked until stable);
//   4. greedy parse with the level's lazy probes (ip+1, ip+2): every lane settles its own position (take my match,
//      or step to a clearly longer one), the scalar unit only hops from one visited match position to the next
//      (one v_readlane per hop); each accepted match is extended backwards over equal preceding bytes down to
//      the previous match's end (the reference's backtrack); a match that reaches past the chunk makes the wave
//      skip whole chunks. Level 6 records the matches instead and runs the optimal parse (zxc_optparse.inc);
//   5. selected lanes emit token / offset / varints (GLO) or 32-bit sequence words (GHI, levels 1-2) at
//      wave-prefix-sum positions; bytes not covered by a match stream into the literal section.
// Finally the literal section is RLE-coded when that is smaller by the reference's margin
// (zxc_compress.c:1270-1534, :1671-1722), the sections are slid together into the reference's layout, or
// the block is stored RAW when that is not smaller. Levels differ in effort (table sizes, chain depth,
// sufficient length, lazy probes: table at the bottom); levels 6-7 code the literal section (7: and the token
// section) with PivCo (zxc_pivco_encode.inc). Output is a valid v8
// block, round-trip-checked by tests/ against the unmodified reference decoder; archive bytes are deterministic.
#include "zxc_experiments.h"  // (first: the gate in front of every experiment switch)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zxc_dev.h"
#include "zxc_encode_levels.h"
#include "zxc_rapidhash.inc"

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// The head table + chain ring are the LDS footprint, i.e. the occupancy (sizes per level: table at the bottom;
// measured trade-offs: profiles/r2o_encode_table_sizes_ab.log).
#define ENC_MARGIN 8u   // the last 8 bytes of a block never start a match (reference ZXC_LZ_SEARCH_MARGIN)
#define ENC_DRY_CHUNKS 8u   // chunks of 64 positions without a sequence before the match finder starts skipping (skip acceleration, below)
// Table entries are the low 16 bits of a position: offsets are < 65536 anyway, so the candidate is
// i - ((i - entry) & 0xFFFF); a stale or never-written entry just names some older position, and
// every candidate is verified against the bytes. Half the LDS of 32-bit entries -> twice the waves.

__device__ __forceinline__ uint32_t e_ld8(const uint8_t* p) { return *p; }
__device__ __forceinline__ uint32_t e_ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t e_ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ v4u e_ld128(const uint8_t* p) { v4u v; __builtin_memcpy(&v, p, 16); return v; }
// 16 bytes at any byte address, fetched DWORD-ALIGNED (four dwords + a fifth, moved into place with v_alignbyte): an unaligned
// 16-byte gather costs the CU's address / data-return pipeline 4 clocks per lane, an aligned one 1.3
// (profiles/r3_vmem_test.log), and the encoder's walk is bound by exactly that pipeline (profiles/r3enc_kprof.txt: TD 75 %
// busy, every ALU under 30 %). `ok` = the 3 bytes in front of p are readable (not the very first bytes of the buffer).
__device__ __forceinline__ v4u e_ld128_al(const uint8_t* p, bool ok) {
    const uint32_t sh = ok ? ((uint32_t)(uintptr_t)p & 3u) : 0u;
    const uint8_t* b = p - sh;
    const v4u a = e_ld128(b);
    const uint32_t e = e_ld32(b + 16);
    v4u r;
    r.x = __builtin_amdgcn_alignbyte(a.y, a.x, sh);
    r.y = __builtin_amdgcn_alignbyte(a.z, a.y, sh);
    r.z = __builtin_amdgcn_alignbyte(a.w, a.z, sh);
    r.w = __builtin_amdgcn_alignbyte(e, a.w, sh);
    return r;
}
__device__ __forceinline__ uint32_t e_uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS traffic between the lanes of the ONE wave of a workgroup: order it (lgkmcnt only). __syncthreads() would also wait
// for every global store and load in flight (vmcnt(0)), i.e. for the token / literal stores of the previous chunk.
__device__ __forceinline__ void enc_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t e_scan_add(uint32_t v) {  // wave inclusive prefix sum (DPP)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t e_wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t t = __shfl_xor(v, d);
        v = t > v ? t : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t varint_len(uint32_t x) { return 1u + (x >= 128u) + (x >= 16384u); }
// prefix varint, docs/FORMAT.md §6 (decoder: src/lib/zxc_decompress.c:51-88)
__device__ __forceinline__ void put_varint(uint8_t* p, uint32_t x) {
    if (x < 128u) {
        p[0] = (uint8_t)x;
    } else if (x < 16384u) {
        p[0] = (uint8_t)(0x80u | (x & 0x3Fu));
        p[1] = (uint8_t)(x >> 6);
    } else {
        p[0] = (uint8_t)(0xC0u | (x & 0x1Fu));
        p[1] = (uint8_t)(x >> 5);
        p[2] = (uint8_t)(x >> 13);
    }
}
// zxc_hash8 (src/lib/zxc_internal.h:1188-1195): block header check byte
__device__ __forceinline__ uint8_t hdr_hash8(uint64_t v) {
    uint64_t h = v ^ 0x9E3779B97F4A7C15ull;
    h ^= h << 13;
    h ^= h >> 7;
    h ^= h << 17;
    return (uint8_t)((h >> 32) ^ h);
}

// wave copy of n bytes, forward, dst below src (regions may overlap that way). Every step moves 1 KiB: the
// wave's loads of a step precede its stores (one instruction each), and a step's stores end below the next
// step's loads because dst < src — no waits beyond the data dependence are needed.
__device__ __forceinline__ void wave_move_down(uint8_t* dst, const uint8_t* src, uint32_t n, int lane) {
    const uint32_t full = n & ~15u;
    for (uint32_t base = 0; base < full; base += 1024u) {
        const uint32_t o = base + 16u * (uint32_t)lane;
        v4u v = {0, 0, 0, 0};
        if (o < full) v = e_ld128(src + o);
        __builtin_amdgcn_wave_barrier();
        if (o < full) __builtin_memcpy(dst + o, &v, 16);
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t tb = 0;
    if (full + (uint32_t)lane < n) tb = src[full + lane];
    __builtin_amdgcn_wave_barrier();
    if (full + (uint32_t)lane < n) dst[full + lane] = (uint8_t)tb;
    __builtin_amdgcn_wave_barrier();
}

// common prefix (bytes, 0..16) of two 16-byte groups (branch-free: the lanes of a wave all take different ways)
__device__ __forceinline__ uint32_t prefix16(uint64_t a_lo, uint64_t a_hi, const v4u c) {
    const uint64_t x = a_lo ^ ((uint64_t)c.x | ((uint64_t)c.y << 32));
    const uint64_t xh = a_hi ^ ((uint64_t)c.z | ((uint64_t)c.w << 32));
    const uint32_t lo = x ? (uint32_t)__builtin_ctzll(x) : 64u;
    const uint32_t hi = xh ? (uint32_t)__builtin_ctzll(xh) : 64u;
    return (x ? lo : 64u + hi) >> 3;
}

#include "zxc_pivco_encode.inc"
#include "zxc_optparse.inc"

// (experiment -DEXP_ENC_CLOCKS, tools/encclk.py: where a chunk's wall time goes. Every phase boundary waits for all outstanding
//  loads, so a phase owns the memory round trips it asked for; sums over all blocks of a launch in shader clocks)
#ifdef EXP_ENC_CLOCKS
__device__ unsigned long long zxc_enc_clk[8];
extern "C" __global__ void zxc_enc_clk_read_kernel(unsigned long long* out) {
    if (threadIdx.x < 8u) { out[threadIdx.x] = zxc_enc_clk[threadIdx.x]; zxc_enc_clk[threadIdx.x] = 0ull; }
}
#define ENC_T(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const uint64_t t_ = __builtin_readcyclecounter(); \
                      clk_acc[k] += t_ - clk_last; clk_last = t_; } while (0)
#elif defined(ASM_MARKERS)  // (design builds: tools/isacount.py counts the instructions between these comments)
#define ENC_T(k) asm volatile("; PHMARK " #k)
#else
#define ENC_T(k) do { } while (0)
#endif

// Slot layout while encoding (stride = 2*block_size + 512 bytes per block):
//   [0,8) block header | [8,20) GLO/GHI header | literals ... | ... staging from block_size + 64: tokens (GLO: 1 B,
//   GHI: 4-byte words), offsets (GLO), extras
// HB = log2(head entries), CWB = log2(chain ring entries) or 0 for "head only". depth / sufficient / lazy: the
// reference's search_depth / sufficient_len / lazy probes (src/lib/zxc_internal.h:965-979), see the table below.
template <uint32_t HSZ, uint32_t CWB, bool GHI, uint32_t NC, uint32_t U>
__device__ __forceinline__ void encode_one_block(const uint8_t* __restrict__ src, uint64_t src_size, uint32_t block_size,
                                                 uint8_t* __restrict__ slots, uint32_t slot_stride,
                                                 uint32_t* __restrict__ sizes, uint32_t n_blocks, uint32_t with_checksum,
                                                 uint32_t depth, uint32_t sufficient, uint32_t lazy, uint32_t dict_size,
                                                 uint8_t* __restrict__ huf_scratch, uint32_t huf) {
    constexpr uint32_t HSIZE = HSZ;  // head entries: any even number (the hash's top bits are scaled onto it), so that the tables can be
                                     // cut to the LDS that buys one more workgroup per CU (160 KiB / 7 = 22.8 KiB)
    constexpr uint32_t CW = CWB ? (1u << CWB) : 1u;
    constexpr uint32_t CWM = CW - 1u;
    constexpr bool DEEP = CWB >= 13u;  // the levels 5-7 entry: the only one that carries the PivCo section encoder
    __shared__ __attribute__((aligned(16))) uint16_t ht[HSIZE];  // head: low 16 bits of the most recent position with this hash
    __shared__ __attribute__((aligned(16))) uint16_t chain[CW];  // chain[q & CWM]: distance from q to the previous position with q's hash (0: none)
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // With a dictionary (reference: zxc_lz_seed_dict + the [dict | block] buffer of zxc_compress_block,
    // src/lib/zxc_dispatch.c:1688-1697) `src` holds one [dict | block] image per block (zxc_prepend_dict_kernel):
    // positions [0, D) are the dictionary — inserted into the tables, never parsed — and the block starts at D.
    const uint32_t D = dict_size;
    const uint8_t* in = src + (uint64_t)b * ((uint64_t)block_size + D);
    const uint64_t remain = src_size - (uint64_t)b * block_size;
    const uint32_t nblk = remain < block_size ? (uint32_t)remain : block_size;  // bytes of the block itself
    const uint32_t n = D + nblk;
    uint8_t* slot = slots + (uint64_t)b * slot_stride;
    // literals are gathered behind room for the widest descriptor set (levels 6-7: lit_comp + tok_comp) and slid down at the end
    const uint32_t lit_base = (DEEP && huf != 0u) ? 28u : 20u;
    uint8_t* lit_out = slot + lit_base;
    const uint32_t max_seq = block_size / 5u + 16u;
    uint8_t* tok_st = slot + block_size + 64u;                // GLO: 1 byte per sequence; GHI: one 32-bit word
    uint8_t* off_st = tok_st + max_seq;                       // GLO: u16 per sequence
    uint8_t* ext_st = GHI ? tok_st + 4u * max_seq : off_st + 2u * max_seq;
    const uint32_t ext_cap = GHI ? block_size / 8u : block_size / 4u;  // staging bound; beyond it the block goes RAW
    const uint32_t esc = GHI ? 255u : 15u;                    // token escape value (LL and ML - 5)

    for (uint32_t i = lane; i < HSIZE / 2u; i += 64u) ((uint32_t*)ht)[i] = 0u;
    if (CWB) for (uint32_t i = lane; i < CW / 2u; i += 64u) ((uint32_t*)chain)[i] = 0u;
    enc_lds_fence();

    // Level 6 (the level table's ZXC_ENC_PARSE_OPTIMAL): the match finder only RECORDS the longest match of every position (mp[], in the block's PivCo scratch, which is
    // idle until the sections are coded); the price-based optimal parse (zxc_optparse.inc) then picks the sequences.
#ifdef EXP_NO_OPTPARSE  // (A/B: level 6 with the lazy parse of round 3)
    const bool OPT = false;
#elif defined(EXP_OPTPARSE_L7)  // (A/B: level 7 with the optimal parse too)
    const bool OPT = DEEP && !GHI && huf != 0u && block_size <= OPT_MAX_BLOCK;
#else
    const bool OPT = DEEP && !GHI && huf != 0u && lazy == ZXC_ENC_PARSE_OPTIMAL && block_size <= OPT_MAX_BLOCK;
#endif
    uint32_t* const mp = OPT ? (uint32_t*)(huf_scratch + (uint64_t)b * 4u * ((uint64_t)block_size + 64u)) : nullptr;
    uint32_t skip_until = 0;  // OPT: positions below it lie strictly inside a match of >= OPT_LONG_SKIP bytes and are not searched
    // Skip acceleration (round 5; reference: step = step_base + (distance from the anchor >> step_shift), src/lib/zxc_compress.c:1176, :1860 —
    // the CPU walks incompressible input in growing steps). Here a chunk is the unit: after ENC_DRY_CHUNKS chunks in a row without a
    // single sequence (512 bytes of literals) only every fourth chunk fetches and compares candidates, the others just enter their
    // positions into the tables (lookup + publish: a quarter of a chunk's cost); the first sequence found ends it. Text never gets there.
    uint32_t dry = 0;  // chunks in a row in which the parse selected nothing
    uint32_t seq_count = 0, lit_count = 0, ext_count = 0, max_off = 0;
    uint32_t pos = D;     // next position the parse will look at
    uint32_t anchor = D;  // end of the last emitted match
    const uint32_t limit = n > ENC_MARGIN + 8u ? n - ENC_MARGIN - 8u : 0u;  // last position that may start a match (exclusive)
    bool overflow = false;
#ifdef EXP_ENC_CLOCKS
    uint64_t clk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t clk_last = __builtin_readcyclecounter();
#endif

    // publish positions of a chunk: chain link = distance to the old head, head = own position
    auto publish = [&](uint32_t i, bool ins, uint32_t h, uint32_t d0) {
        if (ins && CWB) chain[i & CWM] = (uint16_t)d0;
        // head: the highest position of a bucket must win whatever order the hardware applies colliding stores in
        bool want = ins;
        for (;;) {
            if (want) ht[h] = (uint16_t)i;
            enc_lds_fence();
            const uint32_t behind = ins ? ((i - (uint32_t)ht[h]) & 0xFFFFu) : 0u;  // 0: mine is in; 1..63: an earlier lane's
            want = behind != 0u && behind < 64u;
            if (__ballot(want) == 0ull) break;
            enc_lds_fence();
        }
        enc_lds_fence();
    };
    auto hash_of = [&](uint64_t v) -> uint32_t {  // zxc_hash_func, src/lib/zxc_compress.c:45-53: 5-byte / 4-byte variants
        // (top bits of the reference's hash, scaled to the table: for a power of two exactly its `>> (bits - log2 size)`)
        if (GHI) return __umulhi(((uint32_t)v ^ ((uint32_t)v >> 15)) * 0x2D35182Du, HSIZE);
        return __umulhi((uint32_t)(((v & 0xFFFFFFFFFFull) * 0x2545F4914F6CDD1Dull) >> 32), HSIZE);
    };
    // dictionary positions seed the tables (no search, no parse)
    for (uint32_t c0s = 0; c0s < D; c0s += 64u) {
        const uint32_t i = c0s + (uint32_t)lane;
        const bool ins = i < D && i < limit;
        uint32_t h = 0, d0 = 0;
        if (ins) {
            h = hash_of(e_ld64(in + i));
            d0 = (i - (uint32_t)ht[h]) & 0xFFFFu;
            if (d0 > i) d0 = 0;
        }
        enc_lds_fence();
        publish(i, ins, h, d0);
    }

    // The main loop takes ENC_U chunks of 64 positions per iteration (round 3 experiment, kept as a template parameter). A
    // chunk's work is a chain of dependent steps — head lookup, publish, chain links, candidate bytes from memory, parse,
    // emission — and a workgroup owns 20-48 KiB of tables, so only 3-8 waves share a CU and every unit is under half busy
    // (profiles/r3enc_kprof.txt: ~850 instructions, 500 of them scalar, per chunk at ~12 clocks each). With U chunks in flight
    // the candidate loads of all of them are one memory round trip — measured worth nothing (profiles/r3p_encu.log): the
    // wave issues in order and its own dependent ALU / LDS / scalar chains are the time, not the memory round trips.
    // Semantics per chunk are unchanged: its head lookups see every earlier chunk (those of the same iteration included:
    // lookup and publish alternate chunk by chunk), never its own positions.
#ifdef ENC_ALIGNED_CANDIDATES
    const bool lowok = b != 0u || D != 0u || ((uint32_t)(uintptr_t)src & 3u) == 0u;  // (3 readable bytes in front of `in`)
#endif
    uint32_t c0 = D & ~63u;
    v4u v_next[U];  // 16 bytes at every position of the next U chunks
#pragma unroll
    for (uint32_t u = 0; u < U; u++) { const v4u z = {0, 0, 0, 0}; v_next[u] = z; }
    uint32_t c_next = 0xFFFFFFFFu;
    ENC_T(0);  // set-up: tables cleared, dictionary seeded
    while (c0 < n && !overflow) {
        uint32_t iA[U], hA[U], d0A[U], lenA[U], distA[U], bkA[U];
        bool canA[U];
        uint64_t vA[U], vhA[U];
        v4u own1A[U], own2A[U];
        const bool fresh = c_next == c0;
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            iA[u] = c0 + 64u * u + (uint32_t)lane;
            canA[u] = iA[u] < limit && iA[u] >= D && iA[u] >= skip_until;
            if (!fresh && iA[u] < n) v_next[u] = e_ld128(in + iA[u]);  // (a long match skipped ahead: the prefetch was for other chunks)
        }
        // request the following chunks' bytes now; they arrive while these are matched, parsed and emitted
        c_next = c0 + 64u * U;
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            own1A[u] = v_next[u];
            vA[u] = (uint64_t)v_next[u].x | ((uint64_t)v_next[u].y << 32);
            vhA[u] = (uint64_t)v_next[u].z | ((uint64_t)v_next[u].w << 32);
            const uint32_t i2 = c_next + 64u * u + (uint32_t)lane;
            if (i2 < n) v_next[u] = e_ld128(in + i2);  // (every position of the block: its low byte is the literal the emission stores)
            // my own second 16 bytes: the same for every round of the walk (may reach up to 16 bytes past the block: lengths
            // are clamped to it below)
            own2A[u] = e_ld128((canA[u] ? in + iA[u] : in) + 16u);
        }
        // ---- 1. hash -> head candidate, then publish: chunk by chunk, so that a chunk's lookup sees the chunks before it
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            uint32_t h = 0, d0 = 0;
            if (canA[u]) {
                h = hash_of(vA[u]);
                d0 = (iA[u] - (uint32_t)ht[h]) & 0xFFFFu;  // entries hold 16 bits of a position; every candidate is verified
                if (d0 > iA[u]) d0 = 0;                    // (0: none)
            }
            hA[u] = h;
            d0A[u] = d0;
            __builtin_amdgcn_wave_barrier();
            enc_lds_fence();
            publish(iA[u], canA[u], h, d0);
        }
        ENC_T(1);  // own bytes (prefetched) + hash + head lookup + publish
        // ---- 2. chain walks of the U chunks, NC candidates per chunk and round (zxc_lz77_find_best_match :262-440)
        uint32_t triedA[U], dA[U];
#pragma unroll
        // (positions in front of the parse position lie inside the match carried into the chunk: they are in the tables, but the
        //  parse never asks for their matches — no candidates are fetched for them)
#ifdef EXP_ENC_WALK_ALL
        for (uint32_t u = 0; u < U; u++) { lenA[u] = 0; distA[u] = 0; triedA[u] = 0; dA[u] = d0A[u]; }
#else
        for (uint32_t u = 0; u < U; u++) {
            const bool skip = !OPT && dry >= ENC_DRY_CHUNKS && (((c0 >> 6) + u) & 3u) != 0u;  // (wave-uniform)
            lenA[u] = 0; distA[u] = 0; triedA[u] = 0; dA[u] = ((OPT || iA[u] >= pos) && !skip) ? d0A[u] : 0u;
        }
#endif
#ifdef EXP_ENC_NOWALK  // (experiment, wrong output: no candidate is fetched or compared — lookup, publish, parse and emission only)
        for (uint32_t u = 0; u < U; u++) dA[u] = 0;
#endif
        for (;;) {
            bool actA[U];
            uint64_t anyact = 0;
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                actA[u] = dA[u] != 0u && triedA[u] < depth && lenA[u] < sufficient;
                anyact |= __ballot(actA[u]);
            }
            if (anyact == 0ull) break;
            // a link is still in the ring while no newer position has taken its slot (positions up to the end of this
            // iteration's last chunk are in the tables)
            auto next = [&](uint32_t u, uint32_t dk, bool want) -> uint32_t {
                if (!CWB || !want || dk == 0u || dk + 64u * U - (64u * u + (uint32_t)lane) > CW) return 0u;
                const uint32_t dl = chain[(iA[u] - dk) & CWM];
                const uint32_t r = dk + dl;
                return (dl != 0u && r <= 0xFFFFu && r <= iA[u]) ? r : 0u;
            };
            // 32 bytes of every candidate are requested together, for all chunks: most matches end inside them, so a round
            // costs ONE memory round trip; only longer ones enter the extension loop below (requested by every lane — a lane
            // without that candidate re-reads its own position — so that the loads are issued back to back and waited for
            // once). Each candidate's bytes are requested as soon as its distance is known: the further chain links are
            // dependent LDS reads, and their latency then runs under the first candidate's memory round trip.
            uint32_t dkA[U][NC + 1];
            v4u c1A[U][NC];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint8_t* pme = canA[u] ? in + iA[u] : in;
                dkA[u][0] = actA[u] ? dA[u] : 0u;
#pragma unroll
                for (uint32_t k = 0; k < NC; k++) {
#ifdef ENC_ALIGNED_CANDIDATES  // experiment (profiles/r3q_encal.log: +1 % at level 3, -11 % at levels 5-7)
                    c1A[u][k] = e_ld128_al(pme - dkA[u][k], lowok || iA[u] - dkA[u][k] >= 4u);
#else
                    c1A[u][k] = e_ld128(pme - dkA[u][k]);
#endif
                    dkA[u][k + 1] = next(u, dkA[u][k], triedA[u] + k + 1u < depth);
                }
            }
            ENC_T(2);  // chain links + the round's candidate requests, until their bytes are here
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t i = iA[u];
                const uint64_t v = vA[u], vh = vhA[u];
                const uint64_t o2lo = (uint64_t)own2A[u].x | ((uint64_t)own2A[u].y << 32), o2hi = (uint64_t)own2A[u].z | ((uint64_t)own2A[u].w << 32);
                uint32_t mk[NC];
                bool lk[NC];
                bool anylive = false;
#pragma unroll
                for (uint32_t k = 0; k < NC; k++) mk[k] = dkA[u][k] ? prefix16(v, vh, c1A[u][k]) : 0u;
                // the second 16 bytes only for candidates equal over the first 16 (one in five): few lanes, cheap requests
                bool more = false;
#pragma unroll
                for (uint32_t k = 0; k < NC; k++) more |= mk[k] == 16u;
                if (__ballot(more)) {
                    const uint8_t* pme = canA[u] ? in + i : in;
#pragma unroll
                    for (uint32_t k = 0; k < NC; k++) {
                        v4u c2 = {0, 0, 0, 0};
                        if (mk[k] == 16u) c2 = e_ld128(pme - dkA[u][k] + 16u);
                        if (mk[k] == 16u) mk[k] += prefix16(o2lo, o2hi, c2);
                    }
                }
#pragma unroll
                for (uint32_t k = 0; k < NC; k++) {
                    lk[k] = mk[k] == 32u;
                    anylive |= lk[k];
                }
                ENC_T(3);  // first compares + the second 16 bytes
                // candidates still equal after 32 bytes are extended TOGETHER, 16 bytes per step: one request for my own
                // bytes and one per live candidate, all in flight at once, so a step costs one memory round trip however
                // many candidates are still running (the reference extends them one after the other, 8 bytes at a time)
                {
                    uint32_t L = 32u;
                    while (anylive) {
                        if (i + L + 16u > n) {  // block tail: finish bytewise
#pragma unroll
                            for (uint32_t k = 0; k < NC; k++)
                                if (lk[k]) { mk[k] = L; while (i + mk[k] < n && in[i + mk[k]] == in[i - dkA[u][k] + mk[k]]) mk[k]++; lk[k] = false; }
                            break;
                        }
                        const v4u own = e_ld128(in + i + L);
                        v4u xk[NC];
#pragma unroll
                        for (uint32_t k = 0; k < NC; k++) {
                            xk[k] = own;
                            if (lk[k]) xk[k] = e_ld128(in + i - dkA[u][k] + L);
                        }
                        const uint64_t olo = (uint64_t)own.x | ((uint64_t)own.y << 32), ohi = (uint64_t)own.z | ((uint64_t)own.w << 32);
                        anylive = false;
#pragma unroll
                        for (uint32_t k = 0; k < NC; k++) {
                            if (lk[k]) { const uint32_t m = prefix16(olo, ohi, xk[k]); if (m < 16u) { mk[k] = L + m; lk[k] = false; } }
                            anylive |= lk[k];
                        }
                        L += 16u;
                    }
                    // (the tail path above clamps at n; a 16 / 32-byte prefix that straddles the block end is clamped below)
                }
                ENC_T(4);  // extension of candidates equal over 32 bytes
#pragma unroll
                for (uint32_t k = 0; k < NC; k++)
                    if (mk[k] > lenA[u]) { lenA[u] = mk[k]; distA[u] = dkA[u][k]; }   // (first = nearest wins ties: smaller offsets, cheaper tokens)
                triedA[u] += NC;
                dA[u] = dkA[u][NC];
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t i = iA[u];
            if (lenA[u] > n - i) lenA[u] = n - i;
            if (lenA[u] < 5u || !canA[u]) { lenA[u] = 0; distA[u] = 0; }
            // backward extension available at this position: equal bytes just before both sides (<= 16)
            // (with every position inserted, position i-1 finds the same candidate itself, so growing backwards changes
            // next to nothing — identical archive sizes at levels 1-4 in tests/wave_emu — and is kept for the deep levels)
            uint32_t bk = 0;
            if (depth > 8u && lenA[u] && i >= 16u && i - distA[u] >= 16u) {
                const uint64_t y = e_ld64(in + i - 8u) ^ e_ld64(in + i - distA[u] - 8u);
                const uint64_t y2 = e_ld64(in + i - 16u) ^ e_ld64(in + i - distA[u] - 16u);
                bk = y ? (uint32_t)(__builtin_clzll(y) >> 3) : (y2 ? 8u + (uint32_t)(__builtin_clzll(y2) >> 3) : 16u);
            }
            bkA[u] = bk;
        }

        // ---- 4. / 5. parse and emit, chunk by chunk (a chunk an earlier match covers entirely selects and emits nothing)
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t cu = c0 + 64u * u;
            if (cu >= n || overflow) break;
            if (OPT) {  // levels 6-7: record, do not parse (reference: find_best_match inside the DP loop, zxc_compress.c:893-897)
                uint32_t rl = lenA[u];
                uint64_t lm = __ballot(rl >= OPT_LONG_SKIP);
                while (lm) {  // a long match hides the positions inside it (ZXC_OPT_LONG_MATCH_SKIP, :956)
                    const int j = __builtin_ctzll(lm);
                    // (only what the DP can cover: it relaxes lengths up to OPT_LCAP, so the position where a capped match ends is
                    //  searched again and continues the match — hiding the uncapped length left everything behind + OPT_LCAP as literals)
                    const uint32_t jl = (uint32_t)__builtin_amdgcn_readlane((int)rl, j);
                    const uint32_t su = cu + (uint32_t)j + (jl < OPT_LCAP ? jl : OPT_LCAP) - 1u;
                    skip_until = su > skip_until ? su : skip_until;
                    if (iA[u] > cu + (uint32_t)j && iA[u] < skip_until) rl = 0u;
                    const uint32_t upto = skip_until - cu;  // lanes below it are settled
                    lm &= upto >= 64u ? 0ull : ~((1ull << upto) - 1ull);
                }
                if (iA[u] >= D && iA[u] < n) mp[iA[u] - D] = rl >= 5u ? ((rl < 0xFFFFu ? rl : 0xFFFFu) | (distA[u] << 16)) : 0u;
                pos = cu + 64u;
                continue;
            }
#ifdef EXP_ENC_NOPARSE  // (experiment, wrong output: the match finder alone — nothing is parsed or emitted)
            pos = cu + 64u; continue;
#endif
            const uint32_t i = iA[u], len = lenA[u], dist = distA[u], bk = bkA[u];
            const uint64_t v = vA[u];
            ENC_T(5);  // (round bookkeeping, backward extension)
            // ---- 4. parse of the chunk: greedy + the level's lazy probes + backward extension. What the parse does AT a position
            // depends on that position and the two behind it only, so every lane settles its own position first — take my match and
            // go to its end, or step 1 / 2 bytes to a clearly longer one (lazy_len_threshold 128) — and the scalar loop only hops
            // from one visited match position to the next (one v_readlane per hop; round 3 evaluated the probes in the loop, four
            // v_readlane -> s_cmp round trips per sequence: 27 % of the level-3 launch, profiles/r4f_encoder_ablations.log).
            const uint64_t has = __ballot(len >= 5u);  // positions where a match starts
            uint32_t hop = (uint32_t)lane + len;       // where the parse stands after my position, relative to the chunk
            bool take = len >= 5u;
            if (lazy >= 1u) {
                const uint32_t s1 = __shfl(len, (lane + 1) & 63), s2 = __shfl(len, (lane + 2) & 63);  // (every lane takes part: no select around them)
                const uint32_t L1 = lane + 1 < 64 ? s1 : 0u;
                const uint32_t L2 = (lazy >= 2u && lane + 2 < 64) ? s2 : 0u;
                if (take && len < 128u) {
                    if (L1 > len + 1u) { take = false; hop = (uint32_t)lane + 1u; }        // a clearly longer match starts one byte later
                    else if (L2 > len + 2u) { take = false; hop = (uint32_t)lane + 2u; }
                }
            }
            const uint64_t takes = __ballot(take);
            uint64_t sel = 0;
            uint32_t p = pos > cu ? pos - cu : 0u;
            const uint32_t pend = (n - cu < 64u) ? n - cu : 64u;
            {
                // ... and where the next match at or behind a position starts (the chunk's end if none): a hop lands there at once
                const uint64_t mine = has >> lane;
                const uint32_t nm = mine ? (uint32_t)lane + (uint32_t)__builtin_ctzll(mine) : pend;
                const uint32_t landed = __shfl(nm, (int)(hop & 63u));
                hop = hop < 64u ? landed : hop;
                if (p < pend) { const uint64_t ahead = has >> p; p = ahead ? p + (uint32_t)__builtin_ctzll(ahead) : pend; }
                while (p < pend) {
                    sel |= ((takes >> p) & 1ull) << p;
                    p = (uint32_t)__builtin_amdgcn_readlane((int)hop, (int)p);
                }
            }
            const uint32_t next_pos = cu + p;
            ENC_T(6);  // scalar parse

            // ---- 5. emit sequences and literals
            const bool issel = (sel >> lane) & 1ull;
            const uint64_t below = sel & lt_mask;
            const int prevlane = below ? 63 - __builtin_clzll(below) : 0;
            const uint32_t prev_end = __shfl(i + len, prevlane);       // end of the previous selected match (or the carried anchor)
            const uint32_t lit_start = below ? prev_end : anchor;
            // bytes my match grows backwards: not into the previous match (or in front of where the parse stood at the chunk's start)
            const uint32_t floor_abs = below ? prev_end : (pos > cu ? pos : cu);
            const uint32_t ext_v = (issel && bk) ? (bk < i - floor_abs ? bk : i - floor_abs) : 0u;
            const uint32_t mstart = i - ext_v;                          // where my match starts after growing backwards
            const uint32_t ll = issel ? mstart - lit_start : 0u;
            const uint32_t mlm = issel ? len + ext_v - 5u : 0u;
            uint32_t eb = 0;
            if (issel) eb = (ll >= esc ? varint_len(ll - esc) : 0u) + (mlm >= esc ? varint_len(mlm - esc) : 0u);
            const uint32_t eincl = e_scan_add(eb);
            const uint32_t etot = (uint32_t)__builtin_amdgcn_readlane((int)eincl, 63);
            const uint32_t nsel = __popcll(sel);
            if (seq_count + nsel > max_seq || ext_count + etot > ext_cap) { overflow = true; break; }
    #ifndef EXP_ENC_NOSTORE  // (experiment, wrong output: the main loop without its emission stores)
            if (issel) {
                const uint32_t sidx = seq_count + __popcll(below);
                if (GHI) {  // 32-bit word LL(8) | ML-5(8) | offset-1(16), src/lib/zxc_compress.c:1907-1913
                    const uint32_t w = ((ll < 255u ? ll : 255u) << 24) | ((mlm < 255u ? mlm : 255u) << 16) | ((dist - 1u) & 0xFFFFu);
                    __builtin_memcpy(tok_st + 4u * sidx, &w, 4);
                } else {
                    tok_st[sidx] = (uint8_t)(((ll < 15u ? ll : 15u) << 4) | (mlm < 15u ? mlm : 15u));
                    const uint16_t o16 = (uint16_t)(dist - 1u);
                    __builtin_memcpy(off_st + 2u * sidx, &o16, 2);
                }
                uint8_t* e = ext_st + ext_count + eincl - eb;
                if (ll >= esc) { put_varint(e, ll - esc); e += varint_len(ll - esc); }
                if (mlm >= esc) put_varint(e, mlm - esc);
            }
    #endif
            // only the 8-bit / 16-bit offset decision needs the maximum: one ballot instead of a wave reduction
            if (__ballot(issel && dist > 256u)) max_off = 65535u;
            else if (sel && max_off == 0u) max_off = 1u;
            // coverage: selected matches are disjoint and in order. A position is inside a match when it lies before the
            // end of the last selected match at or below it (or of the match carried into the chunk), or at / after the
            // (backwards grown) start of the next selected match above it.
            uint32_t cover_until = pos;
            {
                const uint32_t endv = issel ? i + len : (below ? prev_end : 0u);
                cover_until = endv > cover_until ? endv : cover_until;
            }
            const uint64_t above = sel & ~(lt_mask | (1ull << lane));
            const int nextlane = above ? __builtin_ctzll(above) : 0;
            const uint32_t next_start = __shfl(mstart, nextlane);
            const bool in_next = above != 0ull && i >= next_start;
            // literal = in range, not inside a match, and already passed by the parse
            const bool islit = i < n && i >= cover_until && i < next_pos && !in_next;
            const uint64_t litmask = __ballot(islit);
            // (the byte is already here: low byte of the 16 fetched for this position. No load in the emission: a load whose result a
            // store needs makes the wave wait for every store in front of it, i.e. for this chunk's own sequence stores)
    #ifndef EXP_ENC_NOSTORE
            if (islit) lit_out[lit_count + __popcll(litmask & lt_mask)] = (uint8_t)v;
    #endif
            lit_count += __popcll(litmask);
            seq_count += nsel;
            dry = nsel ? 0u : dry + 1u;  // (also counts the chunks inside a long match: behind one, up to three chunks go unsearched until the next sequence — rare, and three scalar instructions here instead of nine)
            ext_count += etot;
            if (sel) {
                const int last = 63 - __builtin_clzll(sel);
                anchor = __shfl(i + len, last);
            }
            pos = next_pos;
            ENC_T(7);  // emission
        }
#ifdef EXP_ENC_EXTRA_SALU  // experiment only: EXP_ENC_EXTRA_SALU scalar instructions per chunk (is the encoder scalar-issue-bound?)
        {
            uint32_t sd = c0;
#pragma unroll
            for (int q = 0; q < EXP_ENC_EXTRA_SALU; q++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sd) : : "scc");
            asm volatile("" ::"s"(sd));
        }
#endif
#ifdef EXP_ENC_EXTRA_VALU  // experiment only: EXP_ENC_EXTRA_VALU vector instructions per chunk
        {
            uint32_t vd = (uint32_t)lane;
#pragma unroll
            for (int q = 0; q < EXP_ENC_EXTRA_VALU; q++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(vd));
            asm volatile("" ::"v"(vd));
        }
#endif
        // a match reaching past these chunks: skip the chunks it covers entirely
        c0 += 64u * U;
        if (pos > c0) c0 = pos & ~63u;
    }
    __builtin_amdgcn_s_waitcnt(0);
#ifdef EXP_ENC_CLOCKS
    if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&zxc_enc_clk[k], (unsigned long long)clk_acc[k]);
#endif
    if (OPT && !overflow) {
        // ---- levels 6-7: the optimal parse over the recorded matches. The match finder's tables are dead: the chain ring's LDS is
        // the DP window, then the bitmap of match ends; the head table's LDS the histogram, the walk's window, the emitter's list.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // mp[] was written with ordinary stores
        static_assert(!DEEP || (sizeof(uint16_t) * CW >= 8u * OPT_W && sizeof(uint16_t) * HSIZE >= 4u * OPT_W), "DP window / walk window must fit the tables");
        const uint32_t litc = opt_lit_cost(in + D, nblk, (uint32_t*)ht, lane);
        opt_dp(mp, nblk, litc, (unsigned long long*)chain, lane);
        opt_backtrack(mp, nblk, (uint32_t*)ht, (uint32_t*)chain, lane);
        OptOut oo;
        if (!opt_emit(in + D, mp, nblk, (const uint32_t*)chain, (uint32_t*)ht, tok_st, off_st, ext_st, lit_out, max_seq, ext_cap, oo, lane)) overflow = true;
        else { seq_count = oo.seq_count; lit_count = oo.lit_count; ext_count = oo.ext_count; max_off = oo.max_off; }
        __builtin_amdgcn_s_waitcnt(0);
    }

    // ---- RLE literal coding (GLO only): token < 0x80 copies token+1 raw bytes, >= 0x80 repeats the next byte
    // (token & 0x7F) + 4 times (src/lib/zxc_decompress.c:906-975). Chosen when rle_size + 3.125 % of the literal
    // count (below level 6; zxc_internal.h:771-773) beats the raw section. Segmentation as the reference writes it
    // (zxc_compress.c:1671-1722): maximal runs of >= 4 equal bytes become 2-byte run tokens in chunks of <= 131
    // (a remainder of 1-3 bytes is a raw token of its own), everything between them raw tokens of <= 128 bytes.
    uint32_t rle_size = 0;
    bool use_rle = false;
    if (!GHI && !overflow && lit_count >= 64u) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the literal section is read back with cached loads
        uint8_t* rle_out = lit_out + lit_count + 4u;         // (temp behind the literals; moved down when chosen)
        const bool room = lit_base + 4u + 2u * lit_count + 8u <= block_size + 64u;
        // pass 0 sizes, pass 1 writes
        for (uint32_t pass = 0; pass < 2u && room; pass++) {
            if (pass == 1u) {
                const uint32_t tax = (lit_count * 8u) >> 8;  // ZXC_SS_TAX(lit_c, 8)
                if (!(rle_size + tax < lit_count)) break;
                use_rle = true;
            }
            // wave-uniform walk over maximal runs: E bit j of a 64-byte tile = "byte j equals byte j + 1"
            uint32_t w = 0;            // bytes written / sized so far
            uint32_t seg_start = 0;    // first literal byte not yet emitted (start of the pending raw segment)
            uint32_t run_start = 0;    // start of the run of equal bytes the walk is inside
            for (uint32_t t0 = 0; t0 < lit_count; t0 += 64u) {
                const uint32_t j = t0 + (uint32_t)lane;
                const uint32_t b0 = j < lit_count ? e_ld8(lit_out + j) : 0x100u;
                const uint32_t b1 = j + 1u < lit_count ? e_ld8(lit_out + j + 1u) : 0x200u;
                const uint64_t E = __ballot(b0 == b1);
                // run boundaries inside this tile: positions whose byte differs from the next one end a run
                uint64_t ends = ~E;
                const uint32_t tile_n = (lit_count - t0 < 64u) ? lit_count - t0 : 64u;
                if (tile_n < 64u) ends &= (1ull << tile_n) - 1ull;
                while (ends) {
                    const uint32_t k = (uint32_t)__builtin_ctzll(ends);
                    ends &= ends - 1ull;
                    const uint32_t run_end = t0 + k + 1u;           // exclusive
                    const uint32_t run = run_end - run_start;
                    if (run >= 4u) {
                        // raw segment [seg_start, run_start) first
                        uint32_t m = run_start - seg_start;
                        if (pass == 0u) w += m + ((m + 127u) >> 7);
                        else {
                            uint32_t s0 = seg_start;
                            while (m) {
                                const uint32_t c = m > 128u ? 128u : m;
                                if (lane == 0) rle_out[w] = (uint8_t)(c - 1u);
                                for (uint32_t q = lane; q < c; q += 64u) rle_out[w + 1u + q] = lit_out[s0 + q];
                                w += 1u + c; s0 += c; m -= c;
                            }
                        }
                        const uint32_t full = run / 131u, rem = run - full * 131u;
                        if (pass == 0u) w += 2u * full + (rem >= 4u ? 2u : (rem ? 1u + rem : 0u));
                        else {
                            const uint32_t bv = (uint32_t)__builtin_amdgcn_readlane((int)b0, (int)k);
                            for (uint32_t q = lane; q < full; q += 64u) { rle_out[w + 2u * q] = (uint8_t)(0x80u | 127u); rle_out[w + 2u * q + 1u] = (uint8_t)bv; }
                            w += 2u * full;
                            if (rem >= 4u) {
                                if (lane == 0) { rle_out[w] = (uint8_t)(0x80u | (rem - 4u)); rle_out[w + 1u] = (uint8_t)bv; }
                                w += 2u;
                            } else if (rem) {
                                if (lane == 0) rle_out[w] = (uint8_t)(rem - 1u);
                                if ((uint32_t)lane < rem) rle_out[w + 1u + lane] = (uint8_t)bv;
                                w += 1u + rem;
                            }
                        }
                        seg_start = run_end;
                    }
                    run_start = run_end;
                }
            }
            {   // trailing raw segment
                uint32_t m = lit_count - seg_start;
                if (pass == 0u) w += m + ((m + 127u) >> 7);
                else {
                    uint32_t s0 = seg_start;
                    while (m) {
                        const uint32_t c = m > 128u ? 128u : m;
                        if (lane == 0) rle_out[w] = (uint8_t)(c - 1u);
                        for (uint32_t q = lane; q < c; q += 64u) rle_out[w + 1u + q] = lit_out[s0 + q];
                        w += 1u + c; s0 += c; m -= c;
                    }
                }
            }
            if (pass == 0u) rle_size = w;
        }
    }
    // ---- PivCo (Huffman) literal / token coding, levels 6-7 (zxc_pivco_encode.inc). Candidates are priced like the
    // reference does (zxc_compress.c:1536-1626): size + premium * decoded bytes, premium 4/256 for Huffman and 1/256
    // for RLE at these levels (zxc_internal.h:771-795); at least 256 symbols.
    bool lit_huf = false, tok_huf = false;
    uint32_t huf_lit_size = 0, huf_tok_size = 0;
    uint8_t* hs = nullptr;
    if (DEEP && !GHI && huf != 0u && !overflow) {
        PivEnc& PE = *reinterpret_cast<PivEnc*>(ht);  // the match finder's tables are dead now
        static_assert(!DEEP || sizeof(PivEnc) <= sizeof(uint16_t) * HSIZE, "PivEnc must fit the head table");
        const uint32_t hstride = block_size + 64u;
        hs = huf_scratch + (uint64_t)b * 4u * hstride;  // [level buffer A | level buffer B | literal section | token section]
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // literals / tokens were written with ordinary stores
        if (lit_count >= 256u) {
            const uint32_t tax = (lit_count * 4u) >> 8;
            uint32_t best = lit_count;  // raw
            if (use_rle) { const uint32_t j = rle_size + (lit_count >> 8); best = j < best ? j : best; }
            if (best > tax + 129u) {
                huf_lit_size = pivco_encode(lit_out, lit_count, hs + 2u * hstride, best - tax, hs, hs + hstride, PE, lane);
                if (huf_lit_size) { lit_huf = true; use_rle = false; }
            }
        }
        if (huf >= 2u && seq_count >= 256u) {
            const uint32_t tax = (seq_count * 4u) >> 8;
            if (seq_count > tax + 129u) {
                huf_tok_size = pivco_encode(tok_st, seq_count, hs + 3u * hstride, seq_count - tax, hs, hs + hstride, PE, lane);
                tok_huf = huf_tok_size != 0u;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // ---- assemble: [8 B block header][12 B GLO/GHI header][4 B literal descriptor if RLE][literals]
    //      GLO: [tokens][offsets][extras][pad]   GHI: [sequence words][extras][pad]
    const bool off8 = !GHI && max_off <= 256u && max_off != 0u;
    const uint32_t sz_tok = GHI ? 4u * seq_count : (tok_huf ? huf_tok_size : seq_count);
    const uint32_t sz_off = GHI ? 0u : (off8 ? seq_count : 2u * seq_count);
    const uint32_t lit_sec = lit_huf ? huf_lit_size : (use_rle ? rle_size : lit_count);
    const uint32_t desc = ((use_rle || lit_huf) ? 4u : 0u) + (tok_huf ? 4u : 0u);  // lit_comp, then tok_comp (A.2 of SURVEY.md)
    uint32_t behind = sz_tok + sz_off + ext_count;
    const uint32_t pad = behind < 32u ? 32u - behind : 0u;
    const uint32_t payload = 12u + desc + lit_sec + behind + pad;
    if (overflow || 8u + payload >= nblk || nblk < 64u) {
        // RAW block (reference: zxc_encode_block_raw, src/lib/zxc_compress.c:2004-2023)
        for (uint32_t o = 16u * lane; o < nblk; o += 1024u) {
            if (o + 16u <= nblk) { const v4u t = e_ld128(in + D + o); __builtin_memcpy(slot + 8 + o, &t, 16); }
            else for (uint32_t k = o; k < nblk; k++) slot[8 + k] = in[D + k];
        }
        if (lane == 0) {
            uint64_t hv = (uint64_t)0 | ((uint64_t)nblk << 24);  // type 0, flags 0, reserved 0, comp_size le32 @3
            const uint8_t crc = hdr_hash8(hv);
            hv |= (uint64_t)crc << 56;
            __builtin_memcpy(slot, &hv, 8);
        }
        uint32_t total = 8u + nblk;
        if (with_checksum) {  // trailer = checksum of the payload (zxc_compress.c:2060-2071); read back through L2
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const uint32_t ck = wave_checksum32(slot + 8, nblk, lane);
            if (lane == 0) __builtin_memcpy(slot + total, &ck, 4);
            total += 4u;
        }
        if (lane == 0) sizes[b] = total;
        return;
    }
    // a wave copy between regions that do not overlap
    auto wave_copy = [&](uint8_t* dstp, const uint8_t* srcp, uint32_t nbytes) {
        for (uint32_t o = 0; o < nbytes; o += 1024u) {
            const uint32_t q = o + 16u * (uint32_t)lane;
            if (q + 16u <= nbytes) { const v4u t = e_ld128(srcp + q); __builtin_memcpy(dstp + q, &t, 16); }
            else for (uint32_t k = q; k < nbytes; k++) dstp[k] = srcp[k];
        }
    };
    // the literal section goes behind the descriptors: coded in the scratch (PivCo), coded behind the raw literals (RLE),
    // or the raw literals themselves
    if (lit_huf) wave_copy(slot + 20 + desc, hs + 2u * (block_size + 64u), huf_lit_size);
    else if (use_rle) {
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        wave_move_down(slot + 20 + desc, lit_out + lit_count + 4u, rle_size, lane);
    } else if (lit_base != 20u + desc) wave_move_down(slot + 20 + desc, lit_out, lit_count, lane);
    uint8_t* w = slot + 20 + desc + lit_sec;
    if (tok_huf) wave_copy(w, hs + 3u * (block_size + 64u), huf_tok_size);
    else wave_move_down(w, tok_st, sz_tok, lane);
    w += sz_tok;
    if (off8) {
        for (uint32_t s0 = 0; s0 < seq_count; s0 += 64u) {
            const uint32_t sq = s0 + (uint32_t)lane;
            if (sq < seq_count) w[sq] = off_st[2u * sq];
        }
    } else if (!GHI) {
        wave_move_down(w, off_st, 2u * seq_count, lane);
    }
    w += sz_off;
    wave_move_down(w, ext_st, ext_count, lane);
    w += ext_count;
    if ((uint32_t)lane < pad) w[lane] = 0;
    if (lane == 0) {
        uint64_t hv = (GHI ? 2ull : 1ull) | ((uint64_t)payload << 24);  // type 1 = GLO, 2 = GHI
        hv |= (uint64_t)hdr_hash8(hv) << 56;
        __builtin_memcpy(slot, &hv, 8);
        // n_sequences, n_literals, enc_lit (0 raw / 1 RLE / 2 PivCo), enc_tok (0 / 2 PivCo), enc_mlen 0, enc_off
        uint32_t gh[3] = {seq_count, lit_count, (lit_huf ? 2u : (use_rle ? 1u : 0u)) | (tok_huf ? 2u << 8 : 0u) | ((uint32_t)(off8 ? 1u : 0u) << 24)};
        __builtin_memcpy(slot + 8, gh, 12);
        if (use_rle || lit_huf) __builtin_memcpy(slot + 20, &lit_sec, 4);
        if (tok_huf) __builtin_memcpy(slot + 20 + ((use_rle || lit_huf) ? 4 : 0), &huf_tok_size, 4);
    }
    uint32_t total = 8u + payload;
    if (with_checksum) {
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const uint32_t ck = wave_checksum32(slot + 8, payload, lane);
        if (lane == 0) __builtin_memcpy(slot + total, &ck, 4);
        total += 4u;
    }
    if (lane == 0) sizes[b] = total;
}

// Level -> search effort. Reference table (src/lib/zxc_internal.h:965-979): search_depth 3/3/3/3/64/64/128,
// sufficient_len 16/18/16/18/256/256/256, lazy probes 0/0/1/1(+ip+2)/1(+ip+2)/0/0; levels 1-2 emit GHI and use
// the 4-byte hash, levels >= 3 GLO and the 5-byte hash. Here every position is inserted, so chains are denser
// than the CPU's and the deep levels walk fewer links for the same reach; LDS per wave (= occupancy) grows with
// the level: 8 / 12 / 20 / 24 / 48 / 80 / 80 KiB.
//   level   head   chain ring   depth   candidates per round   sufficient   parse     block type
//     1     2^12      -           1          3                    16        greedy    GHI
//     2     2^12     2^11         3          3                    18        greedy    GHI
//     3     2^13     2^11         4          4                    16        lazy 2    GLO
//     4     2^13     2^12         6          6                    18        lazy 2    GLO
//     5     2^13     2^14        18          6                   256        lazy 2    GLO
//     6     2^13     2^15        33          6                   256        optimal   GLO + PivCo literals (zxc_optparse.inc)
//     7     2^13     2^15        66          6                   256        lazy 2    GLO + PivCo literals and tokens
#ifndef ENC_U
#define ENC_U 1u   // chunks of 64 positions in flight per loop iteration (A/B, profiles/r3p_encu.log: 1 / 2 / 3 / 4 all within 2 % at
                   // level 3 — a wave issues in order, only the memory round trips overlap — and 1 keeps the archives of round 2 byte for byte)
#endif
#define ZXC_ENCODE_ENTRY(name, hb, cwb, ghi, waves, nc)                                                                    \
    extern "C" __global__ void __launch_bounds__(64, waves) name(                                                      \
        const uint8_t* __restrict__ src, uint64_t src_size, uint32_t block_size, uint8_t* __restrict__ slots,          \
        uint32_t slot_stride, uint32_t* __restrict__ sizes, uint32_t n_blocks, uint32_t with_checksum, uint32_t depth, \
        uint32_t sufficient, uint32_t lazy, uint32_t dict_size, uint8_t* __restrict__ huf_scratch, uint32_t huf) {     \
        encode_one_block<hb, cwb, ghi, nc, ENC_U>(src, src_size, block_size, slots, slot_stride, sizes, n_blocks, with_checksum,  \
                                       depth, sufficient, lazy, dict_size, huf_scratch, huf);                          \
    }
ZXC_ENCODE_ENTRY(zxc_encode_blocks_kernel_l1, 4096u, 0u, true, 5, 3u)    // level 1 (A/B: one candidate per round instead of three: -3 %)
ZXC_ENCODE_ENTRY(zxc_encode_blocks_kernel_l2, 4096u, 11u, true, 3, 3u)   // level 2
// Level 3 (round 4): head 2^13 + ring 2^11 = 20 KiB -> EIGHT workgroups per CU = two waves on every SIMD instead of six (two SIMDs
// with one wave and nothing to hide its round trips behind): +40 % at the same search effort, for 1.25 % of ratio (the ring's far
// hops); four candidates in ONE round and two lazy probes buy 0.8 % back for 9 % of the time (profiles/r4f_encoder_ablations.log).
#ifndef ENC_L3_HB  // (A/B: tools/build_enc_variant.sh)
#define ENC_L3_HB 13u
#define ENC_L3_CWB 11u
#define ENC_L3_NC 4u
#endif
ZXC_ENCODE_ENTRY(zxc_encode_blocks_kernel_l3, (1u << ENC_L3_HB), ENC_L3_CWB, false, 2, ENC_L3_NC) // level 3
#ifndef ENC_L4_HB  // (A/B: tools/build_enc_variant.sh)
#define ENC_L4_HB 13u
#define ENC_L4_CWB 12u   // 24 KiB -> six per CU: level 4's size bound (1.05 x the reference, today 1.047 x) has nothing to spend on a smaller ring
#define ENC_L4_NC 6u    // (all six candidates in one round: +3 % over two rounds of three, sizes -0.08 %)
#endif
ZXC_ENCODE_ENTRY(zxc_encode_blocks_kernel_l4, (1u << ENC_L4_HB), ENC_L4_CWB, false, 2, ENC_L4_NC) // level 4
#ifndef ENC_L57_HB  // (A/B: tools/build_enc_variant.sh)
#define ENC_L57_HB 13u   // (A/B at level 5 on text, head / ring: 2^14 / 2^14 3.70 GB/s ratio 2.301; 2^13 / 2^14 4.76, 2.287;
#define ENC_L57_CWB 14u  //  2^14 / 2^13 4.99, 2.275; 2^13 / 2^13 6.30, 2.249: the head table is the cheaper one to halve)
#endif
#ifndef ENC_L57_NC
#define ENC_L57_NC 6u   // candidates per round of the deep levels (18 / 33 / 66 per position: half the round trips of 3)
#endif
ZXC_ENCODE_ENTRY(zxc_encode_blocks_kernel_l57, (1u << ENC_L57_HB), ENC_L57_CWB, false, 1, ENC_L57_NC) // level 5
// Levels 6-7 (round 5): a chain ring of 2^15 entries. The reference's chain reaches 65 536 positions back (ZXC_LZ_WINDOW_SIZE,
// src/lib/zxc_common.c:199); with 2^14 the ultra levels came out 2.2-2.5 % larger than the reference's on text, with 2^15 0.8-0.9 %
// (2^16: 0.3 %, but 144 KiB of tables = one workgroup per CU): 80 KiB of tables = two workgroups per CU instead of three —
// the ultra tiers buy ratio with speed (tests/test_wave_emu.py, profiles/r5f_*).
#ifndef ENC_L67_CWB
#define ENC_L67_CWB 15u
#endif
ZXC_ENCODE_ENTRY(zxc_encode_blocks_kernel_l67, (1u << ENC_L57_HB), ENC_L67_CWB, false, 1, ENC_L57_NC) // levels 6-7

// [dict | block b] images for the dictionary path: work + b * (block_size + dict_size)
extern "C" __global__ void __launch_bounds__(64)
zxc_prepend_dict_kernel(const uint8_t* __restrict__ src, uint64_t src_size, uint32_t block_size, const uint8_t* __restrict__ dict,
                        uint32_t dict_size, uint8_t* __restrict__ work, uint32_t n_blocks) {
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const int lane = threadIdx.x;
    uint8_t* w = work + (uint64_t)b * ((uint64_t)block_size + dict_size);
    for (uint32_t o = lane; o < dict_size; o += 64u) w[o] = dict[o];
    const uint64_t remain = src_size - (uint64_t)b * block_size;
    const uint32_t n = remain < block_size ? (uint32_t)remain : block_size;
    const uint8_t* s = src + (uint64_t)b * block_size;
    for (uint32_t o = lane; o < n; o += 64u) w[dict_size + o] = s[o];
}

// offsets[b] = sizes[0] + ... + sizes[b - 1] (one workgroup; a piece of the host API's pipeline is at most a few thousand blocks):
// the compaction below can then follow the encode on its stream without a round trip through the host. Sizes are clamped to
// max_size in place (the encoder never writes more: the host checks the sizes it reads back, this keeps the device in bounds).
extern "C" __global__ void __launch_bounds__(256)
zxc_block_offsets_kernel(uint32_t* __restrict__ sizes, uint64_t* __restrict__ offsets, uint32_t n_blocks, uint32_t max_size) {
    __shared__ uint32_t part[256];
    __shared__ uint64_t carry;
    const uint32_t t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n_blocks; c0 += 256u) {
        const uint32_t b = c0 + t;
        uint32_t v = 0;
        if (b < n_blocks) {
            v = sizes[b];
            if (v > max_size) { v = max_size; sizes[b] = v; }
        }
        part[t] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 256u; d <<= 1) {  // inclusive scan of the chunk
            const uint32_t add = t >= d ? part[t - d] : 0u;
            __syncthreads();
            part[t] += add;
            __syncthreads();
        }
        if (b < n_blocks) offsets[b] = carry + (uint64_t)(part[t] - v);
        __syncthreads();
        if (t == 255u) carry += part[255];
        __syncthreads();
    }
}

// Compaction: block b's bytes [slot, slot+sizes[b]) -> out + offsets[b] (+ optional 4-byte trailer gap).
extern "C" __global__ void __launch_bounds__(64)
zxc_gather_blocks_kernel(const uint8_t* __restrict__ slots, uint32_t slot_stride, const uint32_t* __restrict__ sizes,
                         const uint64_t* __restrict__ offsets, uint8_t* __restrict__ out, uint32_t n_blocks) {
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const int lane = threadIdx.x;
    const uint8_t* s = slots + (uint64_t)b * slot_stride;
    uint8_t* d = out + offsets[b];
    const uint32_t n = sizes[b];
    for (uint32_t o = 16u * lane; o < n; o += 1024u) {
        if (o + 16u <= n) { const v4u t = e_ld128(s + o); __builtin_memcpy(d + o, &t, 16); }
        else for (uint32_t k = o; k < n; k++) d[k] = s[k];
    }
}
