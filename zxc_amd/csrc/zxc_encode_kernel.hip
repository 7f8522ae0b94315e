// zxc_encode_kernel.hip — per-block LZ77 match finder + GLO serialiser for gfx950.
//
// Reference being replaced: zxc_compress_chunk_wrapper (src/lib/zxc_compress.c:2041-2074) →
// zxc_encode_block_glo (:1124-1799) whose hot loop is zxc_lz77_find_best_match (:185-547): one
// thread walks the block, hashing 5 bytes into a 32 K-entry head table + 64 K-entry chain,
// with lazy probes at ip+1/ip+2. That loop is inherently sequential (what gets inserted depends
// on what was matched), so it is NOT reproduced; the wire format only requires what the decoder
// checks (docs/FORMAT.md §5.2): min match 5, 1 <= offset <= 65535 and
// <= bytes produced, token/varint escapes, >= 32 bytes behind the literals, RAW if not smaller.
//
// One wavefront per block. 64 positions per step, one per lane:
//   1. hash of 5 bytes -> LDS table of most recent positions (ds_max_u32: "latest wins",
//      deterministic), candidate verified and extended with 8-byte XOR + ctz compares;
//   2. every lane publishes its position (atomicMax) for later chunks;
//   3. the scalar unit walks the chunk's match lengths with v_readlane: greedy with a one-step
//      lazy probe; a match that reaches past the chunk makes the wave skip whole chunks;
//   4. selected lanes emit token / offset / varints at wave-prefix-sum positions; bytes not
//      covered by a match stream into the literal section, one byte per lane (prefix popcount).
// Finally the sections are slid together into the reference's GLO layout, or the block is stored
// RAW when that is not smaller. Output is a valid v8 block for every level (levels select the
// CPU's parse effort; this matcher has one strategy), round-trip-checked by tests/ against the
// unmodified reference decoder.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zxc_dev.h"
#include "zxc_rapidhash.inc"

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// Hash table size per level class (entries): the table is the LDS footprint, i.e. the occupancy:
// 2^12 -> 8 KiB (20 waves/CU), 2^13 -> 16 KiB (10), 2^14 -> 32 KiB (5). Measured on the text corpus:
// 66 / 42 / 24 GB/s at ratio 1.82 / 1.92 / 1.99 (reference level 3: 1.86).
#define ENC_MARGIN 8u   // the last 8 bytes of a block never start a match (reference ZXC_LZ_SEARCH_MARGIN)
// Table entries are the low 16 bits of a position: offsets are < 65536 anyway, so the candidate is
// i - ((i - entry) & 0xFFFF); a stale or never-written entry just names some older position, and
// every candidate is verified against the bytes. Half the LDS of 32-bit entries -> twice the waves.

__device__ __forceinline__ uint32_t e_ld8(const uint8_t* p) { return *p; }
__device__ __forceinline__ uint64_t e_ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ v4u e_ld128(const uint8_t* p) { v4u v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ uint32_t e_uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

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

// wave copy of n bytes, forward, dst below src (regions may overlap that way): every step
// loads 1 KiB, waits, then stores it
__device__ void wave_move_down(uint8_t* dst, const uint8_t* src, uint32_t n, int lane) {
    const uint32_t full = n & ~15u;
    for (uint32_t o = 16u * lane; o < full; o += 1024u) {
        const v4u v = e_ld128(src + o);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_memcpy(dst + o, &v, 16);
        __builtin_amdgcn_s_waitcnt(0);
    }
    uint32_t tb = 0;
    if (full + (uint32_t)lane < n) tb = src[full + lane];
    __builtin_amdgcn_s_waitcnt(0);
    if (full + (uint32_t)lane < n) dst[full + lane] = (uint8_t)tb;
    __builtin_amdgcn_s_waitcnt(0);
}

// Slot layout while encoding (stride = 2*block_size + 512 bytes per block):
//   [0,8) block header | [8,20) GLO header | literals ... | ... staging: tokens, offsets, extras
template <uint32_t ENC_HBITS>
__device__ __forceinline__ void encode_one_block(const uint8_t* __restrict__ src, uint64_t src_size, uint32_t block_size,
                                                 uint8_t* __restrict__ slots, uint32_t slot_stride,
                                                 uint32_t* __restrict__ sizes, uint32_t n_blocks, uint32_t with_checksum) {
    constexpr uint32_t ENC_HSIZE = 1u << ENC_HBITS;
    __shared__ uint16_t ht[ENC_HSIZE];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint8_t* in = src + (uint64_t)b * block_size;
    const uint64_t remain = src_size - (uint64_t)b * block_size;
    const uint32_t n = remain < block_size ? (uint32_t)remain : block_size;
    uint8_t* slot = slots + (uint64_t)b * slot_stride;
    uint8_t* lit_out = slot + 20;
    const uint32_t max_seq = block_size / 5u + 16u;
    uint8_t* tok_st = slot + block_size + 64u;
    uint8_t* off_st = tok_st + max_seq;       // u16 per sequence
    uint8_t* ext_st = off_st + 2u * max_seq;  // <= 6 bytes per sequence would not fit worst case; bounded below

    for (uint32_t i = lane; i < ENC_HSIZE / 2u; i += 64u) ((uint32_t*)ht)[i] = 0u;
    __syncthreads();

    uint32_t seq_count = 0, lit_count = 0, ext_count = 0, max_off = 0;
    uint32_t pos = 0;     // next position the parse will look at
    uint32_t anchor = 0;  // end of the last emitted match
    const uint32_t limit = n > ENC_MARGIN + 8u ? n - ENC_MARGIN - 8u : 0u;  // last position that may start a match (exclusive)
    const uint32_t ext_cap = block_size / 4u;                                // staging bound; beyond it the block goes RAW
    bool overflow = false;

    uint32_t c0 = 0;
    v4u v_next = {0, 0, 0, 0};  // 16 bytes at every position of the next chunk
    if ((uint32_t)lane < limit) v_next = e_ld128(in + lane);
    uint32_t c_next = 0;
    while (c0 < n) {
        const uint32_t i = c0 + (uint32_t)lane;
        // ---- 1. candidate + verified length for every position of the chunk
        uint32_t len = 0, cpos = 0;
        uint64_t v = 0, vh = 0;
        const bool can = i < limit;
        uint32_t h = 0;
        if (c_next != c0 && can) v_next = e_ld128(in + i);  // (a long match skipped ahead: the prefetch was for another chunk)
        {   // request the following chunk's bytes now; they arrive while this chunk is matched, parsed and emitted
            v = (uint64_t)v_next.x | ((uint64_t)v_next.y << 32);
            vh = (uint64_t)v_next.z | ((uint64_t)v_next.w << 32);
            c_next = c0 + 64u;
            const uint32_t i2 = c_next + (uint32_t)lane;
            if (i2 < limit) v_next = e_ld128(in + i2);
        }
        if (can) {
            h = (uint32_t)(((v & 0xFFFFFFFFFFull) * 0x9E3779B185EBCA87ull) >> (64u - ENC_HBITS));
            const uint32_t dist = (i - (uint32_t)ht[h]) & 0xFFFFu;
            {
                cpos = i - dist;
                if (dist != 0u && dist <= i) {
                    // 16 bytes of both sides in one round trip: most matches end inside them
                    const v4u cv = e_ld128(in + cpos);
                    const uint64_t x = v ^ ((uint64_t)cv.x | ((uint64_t)cv.y << 32));
                    const uint64_t xh = vh ^ ((uint64_t)cv.z | ((uint64_t)cv.w << 32));
                    if (x == 0 && xh != 0) {
                        len = 8u + (uint32_t)(__builtin_ctzll(xh) >> 3);
                    } else if (x == 0) {
                        len = 16;
                        while (i + len + 8u <= n) {  // extend, 8 bytes at a time
                            const uint64_t y = e_ld64(in + i + len) ^ e_ld64(in + cpos + len);
                            if (y) { len += (uint32_t)(__builtin_ctzll(y) >> 3); break; }
                            len += 8u;
                        }
                        if (len > n - i) len = n - i;
                        while (i + len < n && in[i + len] == in[cpos + len]) len++;
                    } else {
                        len = (uint32_t)(__builtin_ctzll(x) >> 3);
                    }
                    if (len < 5u) len = 0;
                }
            }
        }
        // ---- 2. publish this chunk's positions (most recent wins)
        if (can) ht[h] = (uint16_t)i;
        __syncthreads();

        // ---- 3. scalar parse of the chunk
        uint64_t sel = 0;
        uint32_t p = pos > c0 ? pos - c0 : 0u;
        const uint64_t has = __ballot(len >= 5u);  // positions where a match starts
        const uint32_t pend = (n - c0 < 64u) ? n - c0 : 64u;
        while (p < pend) {
            const uint64_t ahead = has >> p;
            if (ahead == 0ull) { p = pend; break; }  // nothing left in this chunk: all literals
            p += (uint32_t)__builtin_ctzll(ahead);
            if (p >= pend) { p = pend; break; }
            const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)p);
            const uint32_t L1 = (p + 1u < 64u) ? (uint32_t)__builtin_amdgcn_readlane((int)len, (int)(p + 1u)) : 0u;
            if (L1 > L + 1u) { p++; continue; }  // lazy: a clearly longer match starts one byte later
            sel |= 1ull << p;
            p += L;
        }
        const uint32_t next_pos = c0 + p;

        // ---- 4. emit sequences and literals
        const bool issel = (sel >> lane) & 1ull;
        // end of the previous selected match (or the carried anchor)
        const uint64_t below = sel & lt_mask;
        const int prevlane = below ? 63 - __builtin_clzll(below) : 0;
        const uint32_t prev_end = __shfl(i + len, prevlane);
        const uint32_t lit_start = below ? prev_end : anchor;
        const uint32_t ll = issel ? i - lit_start : 0u;
        const uint32_t mlm = issel ? len - 5u : 0u;
        const uint32_t off = i - cpos;
        uint32_t eb = 0;
        if (issel) eb = (ll >= 15u ? varint_len(ll - 15u) : 0u) + (mlm >= 15u ? varint_len(mlm - 15u) : 0u);
        const uint32_t eincl = e_scan_add(eb);
        const uint32_t etot = (uint32_t)__builtin_amdgcn_readlane((int)eincl, 63);
        const uint32_t nsel = __popcll(sel);
        if (seq_count + nsel > max_seq || ext_count + etot > ext_cap) { overflow = true; break; }
        if (issel) {
            const uint32_t sidx = seq_count + __popcll(below);
            tok_st[sidx] = (uint8_t)(((ll < 15u ? ll : 15u) << 4) | (mlm < 15u ? mlm : 15u));
            const uint16_t o16 = (uint16_t)(off - 1u);
            __builtin_memcpy(off_st + 2u * sidx, &o16, 2);
            uint8_t* e = ext_st + ext_count + eincl - eb;
            if (ll >= 15u) { put_varint(e, ll - 15u); e += varint_len(ll - 15u); }
            if (mlm >= 15u) put_varint(e, mlm - 15u);
        }
        // only the 8-bit / 16-bit offset decision needs the maximum: one ballot instead of a wave reduction
        if (__ballot(issel && off > 256u)) max_off = 65535u;
        else if (sel && max_off == 0u) max_off = 1u;
        // coverage: selected matches are disjoint and in order, so a position is covered exactly when it lies
        // before the end of the last selected match at or below it (or of the match carried into the chunk)
        uint32_t cover_until = pos;  // positions < pos are covered by a match that started earlier
        {
            const uint32_t endv = issel ? i + len : (below ? prev_end : 0u);
            cover_until = endv > cover_until ? endv : cover_until;
        }
        // literal = in range, not inside a match, and already passed by the parse
        const bool islit = i < n && i >= cover_until && i < next_pos;
        const uint64_t litmask = __ballot(islit);
        // (the byte is already here: low byte of the 16 fetched for this position; only the block's last 16
        // positions, which never start a match, were not fetched)
        if (islit) lit_out[lit_count + __popcll(litmask & lt_mask)] = can ? (uint8_t)v : (uint8_t)e_ld8(in + i);
        lit_count += __popcll(litmask);
        seq_count += nsel;
        ext_count += etot;
        if (sel) {
            const int last = 63 - __builtin_clzll(sel);
            anchor = __shfl(i + len, last);
        }
        pos = next_pos;
        // a match reaching past this chunk: skip the chunks it covers entirely
        c0 += 64u;
        if (pos > c0) c0 = pos & ~63u;
    }
    __builtin_amdgcn_s_waitcnt(0);

    // ---- assemble: [8 B block header][12 B GLO header][literals][tokens][offsets][extras][pad]
    const bool off8 = max_off <= 256u && max_off != 0u;
    const uint32_t sz_off = off8 ? seq_count : 2u * seq_count;
    uint32_t behind = seq_count + sz_off + ext_count;
    const uint32_t pad = behind < 32u ? 32u - behind : 0u;
    const uint32_t payload = 12u + lit_count + behind + pad;
    if (overflow || 8u + payload >= n || n < 64u) {
        // RAW block (reference: zxc_encode_block_raw, src/lib/zxc_compress.c:2004-2023)
        for (uint32_t o = 16u * lane; o < n; o += 1024u) {
            if (o + 16u <= n) { const v4u t = e_ld128(in + o); __builtin_memcpy(slot + 8 + o, &t, 16); }
            else for (uint32_t k = o; k < n; k++) slot[8 + k] = in[k];
        }
        if (lane == 0) {
            uint64_t hv = (uint64_t)0 | ((uint64_t)n << 24);  // type 0, flags 0, reserved 0, comp_size le32 @3
            const uint8_t crc = hdr_hash8(hv);
            hv |= (uint64_t)crc << 56;
            __builtin_memcpy(slot, &hv, 8);
        }
        uint32_t total = 8u + n;
        if (with_checksum) {  // trailer = checksum of the payload (zxc_compress.c:2060-2071); read back through L2
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const uint32_t ck = wave_checksum32(slot + 8, n, lane);
            if (lane == 0) __builtin_memcpy(slot + total, &ck, 4);
            total += 4u;
        }
        if (lane == 0) sizes[b] = total;
        return;
    }
    uint8_t* w = slot + 20 + lit_count;
    wave_move_down(w, tok_st, seq_count, lane);
    w += seq_count;
    if (off8) {
        for (uint32_t s = lane; s < seq_count; s += 64u) w[s] = off_st[2u * s];
        __builtin_amdgcn_s_waitcnt(0);
    } else {
        wave_move_down(w, off_st, 2u * seq_count, lane);
    }
    w += sz_off;
    wave_move_down(w, ext_st, ext_count, lane);
    w += ext_count;
    if ((uint32_t)lane < pad) w[lane] = 0;
    if (lane == 0) {
        uint64_t hv = 1ull | ((uint64_t)payload << 24);  // type 1 = GLO
        hv |= (uint64_t)hdr_hash8(hv) << 56;
        __builtin_memcpy(slot, &hv, 8);
        uint32_t gh[3] = {seq_count, lit_count, (uint32_t)(off8 ? 1u : 0u) << 24};  // enc_lit 0, enc_tok 0, enc_mlen 0, enc_off
        __builtin_memcpy(slot + 8, gh, 12);
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

#define ZXC_ENCODE_ENTRY(name, bits)                                                                                   \
    extern "C" __global__ void __launch_bounds__(64) name(                                                             \
        const uint8_t* __restrict__ src, uint64_t src_size, uint32_t block_size, uint8_t* __restrict__ slots,          \
        uint32_t slot_stride, uint32_t* __restrict__ sizes, uint32_t n_blocks, uint32_t with_checksum) {               \
        encode_one_block<bits>(src, src_size, block_size, slots, slot_stride, sizes, n_blocks, with_checksum);         \
    }
ZXC_ENCODE_ENTRY(zxc_encode_blocks_kernel_h12, 12u)  // levels 1-2
ZXC_ENCODE_ENTRY(zxc_encode_blocks_kernel_h13, 13u)  // levels 3-4
ZXC_ENCODE_ENTRY(zxc_encode_blocks_kernel_h14, 14u)  // levels 5-7

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
