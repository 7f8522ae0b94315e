// wave_emu.h — lock-step emulation of the gfx950 wavefronts of one workgroup on the CPU (test infrastructure).
//
// The device sources under zxc_amd/csrc are compiled unchanged for x86 with this header standing
// in for <hip/hip_runtime.h>: each of the 64 lanes of a wavefront is a fiber (ucontext), and every
// cross-lane operation (DPP, ds_bpermute / __shfl, ballot, readlane, fences and waits that order
// memory between lanes) is a rendezvous of all live lanes. Between two rendezvous the lanes run one
// after the other, so code that needs an LDS / memory fence between a write of one lane and the read
// of another and does not have one fails here as well. A rendezvous reached by lanes coming from
// different call sites (a cross-lane operation inside divergent control flow) aborts with a message.
// Purpose: check the kernel LOGIC against the oracle on machines without a GPU. It says nothing
// about timing, s_waitcnt placement or hardware memory-model behaviour.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <type_traits>

namespace emu {
constexpr int W = 64;
struct Idx { unsigned x, y, z; };
// rendezvous: every live lane contributes v; returns the 64 contributed values (dead lanes: 0)
const uint64_t* sync(uint64_t v, int tag, uint64_t* active);
int lane();
Idx tid();
Idx bid();
Idx gdim();
[[noreturn]] void fail(const char* what, int tag);
void note_exit();

inline uint32_t readfirstlane(uint32_t v, int tag) {
    uint64_t act; const uint64_t* s = sync(v, tag, &act);
    return (uint32_t)s[__builtin_ctzll(act)];
}
inline uint32_t readlane(uint32_t v, int l, int tag) {
    uint64_t act; const uint64_t* s = sync(v, tag, &act);
    return (uint32_t)s[l & 63];
}
inline uint32_t shfl(uint32_t v, int l, int tag) {  // ds_bpermute_b32
    uint64_t act; const uint64_t* s = sync(v, tag, &act);
    return (uint32_t)s[l & 63];
}
inline uint32_t shfl_up(uint32_t v, unsigned d, int tag) {
    uint64_t act; const uint64_t* s = sync(v, tag, &act);
    const int me = lane();
    return me >= (int)d ? (uint32_t)s[me - d] : v;
}
inline uint32_t shfl_xor(uint32_t v, int m, int tag) {
    uint64_t act; const uint64_t* s = sync(v, tag, &act);
    return (uint32_t)s[(lane() ^ m) & 63];
}
inline uint64_t ballot(bool p, int tag) {
    uint64_t act; const uint64_t* s = sync(p ? 1u : 0u, tag, &act);
    uint64_t m = 0;
    for (int i = 0; i < W; i++) if (((act >> i) & 1) && s[i]) m |= 1ull << i;
    return m;
}
inline void barrier(int tag) { uint64_t act; (void)sync(0, tag, &act); }
void wg_barrier(int tag);  // s_barrier: every live lane of the workgroup (wave_emu.cpp)
// v_mov_b32_dpp semantics for the controls the kernels use (gfx9 DPP): row_shr:n, row_shl:n, row_ror:n,
// wave_shr:1 / wave_shl:1, row_bcast:15, row_bcast:31, quad_perm, row_mirror, row_half_mirror.
inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, int tag) {
    uint64_t act; const uint64_t* s = sync((uint32_t)src, tag, &act);
    const int me = lane(), row = me >> 4, bank = (me & 15) >> 2, k = me & 15;
    if (!((row_mask >> row) & 1) || !((bank_mask >> bank) & 1)) return old;
    int from = -1;
    if (ctrl >= 0x000 && ctrl <= 0x0FF) from = (me & ~3) | ((ctrl >> (2 * (me & 3))) & 3);           // quad_perm
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; if (k + n < 16) from = me + n; }  // row_shl
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; if (k >= n) from = me - n; }      // row_shr
    else if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; from = (me & ~15) | ((k - n) & 15); }  // row_ror
    else if (ctrl == 0x130) { if (me < 63) from = me + 1; }   // wave_shl:1
    else if (ctrl == 0x138) { if (me > 0) from = me - 1; }    // wave_shr:1
    else if (ctrl == 0x140) from = (me & ~15) | (15 - k);     // row_mirror
    else if (ctrl == 0x141) from = (me & ~7) | (7 - (me & 7)); // row_half_mirror
    else if (ctrl == 0x142) { if (row >= 1) from = row * 16 - 1; }            // row_bcast:15
    else if (ctrl == 0x143) { if (row >= 2) from = 31; }                       // row_bcast:31
    else fail("unsupported dpp_ctrl", ctrl);
    if (from < 0 || !((act >> from) & 1)) return bound_ctrl ? 0 : old;
    return (int)(uint32_t)s[from];
}
inline uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (sh & 3u)));
}
inline uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {  // v_alignbit_b32
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u));
}
inline uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) {  // v_perm_b32
    const uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t c = (sel >> (8 * i)) & 255u;
        uint32_t b;
        if (c <= 7) b = (uint32_t)(src >> (8 * c)) & 255u;
        else if (c == 8) b = ((s1 >> 15) & 1) ? 255u : 0u;
        else if (c == 9) b = ((s1 >> 31) & 1) ? 255u : 0u;
        else if (c == 10) b = ((s0 >> 15) & 1) ? 255u : 0u;
        else if (c == 11) b = ((s0 >> 31) & 1) ? 255u : 0u;
        else if (c == 12) b = 0;
        else b = 255u;
        r |= b << (8 * i);
    }
    return r;
}
inline uint32_t mbcnt(uint32_t mask, uint32_t add, bool hi) {
    const int l = lane();
    uint32_t m;
    if (!hi) m = l >= 32 ? 0xFFFFFFFFu : ((1u << l) - 1u);
    else m = l <= 32 ? 0u : ((1u << (l - 32)) - 1u);
    return add + (uint32_t)__builtin_popcount(mask & m);
}
template <typename T> inline T atomic_cas(T* p, T cmp, T val) { T o = *p; if (o == cmp) *p = val; return o; }
template <typename T> inline T atomic_add(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomic_max(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomic_or(T* p, T v) { T o = *p; *p = o | v; return o; }
uint64_t clock64();
template <typename T> inline T ntload(const void* p) { T v; memcpy(&v, p, sizeof(T)); return v; }  // device loads need only dword alignment
}  // namespace emu

// ---------------------------------------------------------------- HIP surface used by the kernels
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static __attribute__((section("emu_lds")))
#define threadIdx (emu::tid())
#define blockIdx (emu::bid())
#define gridDim (emu::gdim())
#define warpSize 64

#define __builtin_amdgcn_readfirstlane(v) ((int)emu::readfirstlane((uint32_t)(v), __LINE__))
#define __builtin_amdgcn_readlane(v, l) ((int)emu::readlane((uint32_t)(v), (l), __LINE__))
#define __builtin_amdgcn_update_dpp(o, s, c, rm, bm, bc) emu::update_dpp((o), (s), (c), (rm), (bm), (bc), __LINE__)
#define __builtin_amdgcn_mov_dpp(s, c, rm, bm, bc) emu::update_dpp(0, (s), (c), (rm), (bm), (bc), __LINE__)
#define __builtin_amdgcn_alignbyte(h, l, s) emu::alignbyte((h), (l), (s))
#define __builtin_amdgcn_perm(a, b, s) emu::perm((a), (b), (s))
#define __builtin_amdgcn_alignbit(h, l, s) emu::alignbit((h), (l), (s))
#define __builtin_amdgcn_mbcnt_lo(m, a) emu::mbcnt((m), (a), false)
#define __builtin_amdgcn_mbcnt_hi(m, a) emu::mbcnt((m), (a), true)
#define __builtin_amdgcn_ds_bpermute(addr, v) ((int)emu::shfl((uint32_t)(v), (int)((addr) >> 2), __LINE__))
// fences, waits and barriers order memory between the lanes of the wave: rendezvous
#define __builtin_amdgcn_fence(order, scope) emu::barrier(__LINE__)
#define __builtin_amdgcn_wave_barrier() emu::barrier(__LINE__)
#define __builtin_amdgcn_s_waitcnt(n) emu::barrier(__LINE__)
#define __builtin_amdgcn_s_barrier() emu::wg_barrier(__LINE__)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_nontemporal_load(p) emu::ntload<std::remove_cv_t<std::remove_reference_t<decltype(*(p))>>>((const void*)(p))
#define address_space(n)   /* __attribute__((address_space(1))) pointers are ordinary pointers here */
#define __syncthreads() emu::wg_barrier(__LINE__)
#define __shfl(v, l) emu::shfl((uint32_t)(v), (int)(l), __LINE__)
#define __shfl_up(v, d) emu::shfl_up((uint32_t)(v), (unsigned)(d), __LINE__)
#define __shfl_xor(v, m) emu::shfl_xor((uint32_t)(v), (int)(m), __LINE__)
#define __ballot(p) emu::ballot((p), __LINE__)
#define __popc(x) __builtin_popcount(x)
#define __popcll(x) __builtin_popcountll(x)
#define __ffsll(x) __builtin_ffsll((long long)(x))
#define __ffs(x) __builtin_ffs((int)(x))
#define __umul24(a, b) (((uint32_t)(a) & 0xFFFFFFu) * ((uint32_t)(b) & 0xFFFFFFu))
#define __umulhi(a, b) ((uint32_t)(((uint64_t)(uint32_t)(a) * (uint64_t)(uint32_t)(b)) >> 32))
#define __umul64hi(a, b) ((uint64_t)(((unsigned __int128)(a) * (unsigned __int128)(b)) >> 64))
inline uint32_t atomicCAS(uint32_t* p, uint32_t c, uint32_t v) { return emu::atomic_cas(p, c, v); }
inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { return emu::atomic_add(p, v); }
inline uint32_t atomicMax(uint32_t* p, uint32_t v) { return emu::atomic_max(p, v); }
inline uint32_t atomicOr(uint32_t* p, uint32_t v) { return emu::atomic_or(p, v); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return emu::atomic_add(p, v); }
inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; if (v < o) *p = v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_load(p, order, scope) (*(p))
#define wall_clock64() emu::clock64()
