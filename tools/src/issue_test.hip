// Micro-benchmark: what does one wave64 instruction cost on a gfx950 SIMD, per instruction class, at 1 / 2 / 5 / 8
// waves per SIMD?  Answers VERDICT r2 weak #1: "clocks per wave64 v_add_u32 / v_alignbyte / v_and_or / v_cndmask at
// 5 waves/SIMD; state VALU-busy as one number".
// Each kernel runs ITERS x 64 copies of a 4-instruction body on 4 independent register chains (so dependent-issue
// latency does not bound it once >= 2 waves share the SIMD) and reports
//   clk/instr/SIMD = (shader clocks of the slowest wave, s_memtime) / (instructions issued by all waves of that SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/issue_test tools/src/issue_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define BODY_VADD   "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0"
#define BODY_VXOR   "v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0"
#define BODY_ALIGN  "v_alignbyte_b32 %0, %0, %1, 1\n v_alignbyte_b32 %1, %1, %2, 2\n v_alignbyte_b32 %2, %2, %3, 3\n v_alignbyte_b32 %3, %3, %0, 1"
#define BODY_ALIGNV "v_alignbyte_b32 %0, %0, %1, %2\n v_alignbyte_b32 %1, %1, %2, %3\n v_alignbyte_b32 %2, %2, %3, %0\n v_alignbyte_b32 %3, %3, %0, %1"
#define BODY_ANDOR  "v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %0\n v_and_or_b32 %3, %3, %0, %1"
#define BODY_CNDM   "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc"
#define BODY_CMP    "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0"
#define BODY_PERM   "v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1"
#define BODY_BFE    "v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9"
#define BODY_LSHL64 "v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1"
#define BODY_MULLO  "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0"
#define BODY_MAD24  "v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1"
#define BODY_DPP    "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
#define BODY_DPPBC  "v_add_u32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %1, %1, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_u32_dpp %2, %2, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %3, %3, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
#define BODY_MBCNT  "v_mbcnt_lo_u32_b32 %0, %1, %0\n v_mbcnt_hi_u32_b32 %1, %2, %1\n v_mbcnt_lo_u32_b32 %2, %3, %2\n v_mbcnt_hi_u32_b32 %3, %0, %3"
#define BODY_BCNT   "v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %1, %2, %1\n v_bcnt_u32_b32 %2, %3, %2\n v_bcnt_u32_b32 %3, %0, %3"
#define BODY_PKADD  "v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %3, %3, %0"
#define BODY_VAND   "v_and_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_or_b32 %3, %3, %0"
#define BODY_VSHL   "v_lshlrev_b32 %0, %0, %1\n v_lshrrev_b32 %1, %1, %2\n v_lshlrev_b32 %2, %2, %3\n v_lshrrev_b32 %3, %3, %0"
#define BODY_VSUB   "v_sub_u32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_max_u32 %2, %2, %3\n v_subrev_u32 %3, %3, %0"
#define BODY_VMOV   "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0"
#define BODY_VADDLIT "v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3"
#define BODY_VADDK  "v_add_u32 %0, 17, %0\n v_add_u32 %1, 17, %1\n v_add_u32 %2, 17, %2\n v_add_u32 %3, 17, %3"
#define BODY_VADDCO "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_add_co_u32 %2, vcc, %2, %3\n v_addc_co_u32 %3, vcc, %3, %0, vcc"
#define BODY_LSHLADD "v_lshl_add_u32 %0, %0, 2, %1\n v_lshl_add_u32 %1, %1, 2, %2\n v_lshl_add_u32 %2, %2, 2, %3\n v_lshl_add_u32 %3, %3, 2, %0"
#define BODY_ADD3   "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1"
#define BODY_CNDS   "v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %1, %1, %2, s[20:21]\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cndmask_b32 %3, %3, %0, s[20:21]"
#define BODY_CMPCND "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %3, %3, %0, vcc"
#define BODY_CMPS   "v_cmp_lt_u32 s[20:21], %0, %1\n v_cmp_lt_u32 s[22:23], %1, %2\n v_cmp_lt_u32 s[20:21], %2, %3\n v_cmp_lt_u32 s[22:23], %3, %0"
#define BODY_MOVDPP "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf"
#define BODY_SDWA   "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD"
#define BODY_RFL    "v_readfirstlane_b32 %0, %4\n v_readfirstlane_b32 %1, %5\n v_readfirstlane_b32 %2, %6\n v_readfirstlane_b32 %3, %7"
#define BODY_SADD   "s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_xor_b32 %3, %3, %0"

enum { K_VADD, K_VXOR, K_ALIGN, K_ALIGNV, K_ANDOR, K_CNDM, K_CMP, K_PERM, K_BFE, K_LSHL64, K_MULLO, K_MAD24, K_DPP, K_DPPBC,
       K_MBCNT, K_BCNT, K_PKADD, K_VAND, K_VSHL, K_VSUB, K_VMOV, K_VADDLIT, K_VADDK, K_VADDCO, K_LSHLADD, K_ADD3, K_CNDS, K_CMPCND, K_CMPS, K_MOVDPP, K_SDWA, K_RFL, K_DSREADU8, K_DSWRITEB8, K_DSREAD64, K_DSOR_RAND, K_DSREAD_RAND, K_SADD, K_READLANE, K_BPERM, K_DSREAD, K_DSREAD128, K_DSOR, K_DSWRITE, K_VALU_SALU, K_BALLOT, K_N };
static const char* NAMES[K_N] = {"v_add_u32", "v_xor_b32", "v_alignbyte(imm)", "v_alignbyte(vgpr)", "v_and_or_b32", "v_cndmask_b32",
    "v_cmp_lt_u32", "v_perm_b32", "v_bfe_u32", "v_lshlrev_b64", "v_mul_lo_u32", "v_mad_u32_u24", "v_add_u32 dpp row_shr",
    "v_add_u32 dpp row_bcast", "v_mbcnt", "v_bcnt_u32", "v_pk_add_u16", "v_and/v_or", "v_lshl/v_lshr", "v_sub/min/max", "v_mov_b32", "v_add_u32 (32-bit literal)", "v_add_u32 (inline const)", "v_add_co/v_addc_co", "v_lshl_add_u32", "v_add3_u32", "v_cndmask (sgpr pair mask)", "v_cmp vcc + v_cndmask vcc", "v_cmp -> sgpr pair", "v_mov_b32 dpp row_shr", "v_add_u32 sdwa", "v_readfirstlane", "ds_read_u8 (lane-linear)", "ds_write_b8 (lane-linear)", "ds_read_b64 (lane-linear)", "ds_or_b32 (random dword)", "ds_read_b32 (random dword)", "s_add/s_xor (SALU)", "v_readlane_b32 (sgpr idx)",
    "ds_bpermute_b32", "ds_read_b32 (lane-linear)", "ds_read_b128 (lane-linear)", "ds_or_b32 (lane-linear)", "ds_write_b32 (lane-linear)",
    "2 v_add + 2 s_add interleaved", "v_cmp + s_and (ballot-like)"};

template <int KIND>
__global__ void __launch_bounds__(64) k(uint32_t* out, uint64_t* clk, int iters, uint32_t seed) {
    __shared__ uint32_t lds[1024 + 8];
    uint32_t v0 = threadIdx.x + seed, v1 = v0 * 3u + 1u, v2 = (v0 ^ 5u) & 3u, v3 = (v0 + 7u) & 3u;
    uint64_t w0 = v0, w1 = v1;
    uint32_t s0 = seed, s1 = seed * 3u, s2 = seed ^ 9u, s3 = seed + 11u;
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = i;
    __syncthreads();
    const uint32_t la = (threadIdx.x * 4u) + (uint32_t)(size_t)lds, la16 = (threadIdx.x * 16u) + (uint32_t)(size_t)lds;
    // a per-lane pseudo-random 4-aligned address within 4 KiB - 16 (the ring's access pattern: every lane somewhere else)
    const uint32_t lrand = (uint32_t)(size_t)lds + (((threadIdx.x * 2654435761u) >> 7) % 1020u) * 4u;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++) {
            if (KIND == K_VADD) asm volatile(BODY_VADD : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_VXOR) asm volatile(BODY_VXOR : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_ALIGN) asm volatile(BODY_ALIGN : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_ALIGNV) asm volatile(BODY_ALIGNV : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_ANDOR) asm volatile(BODY_ANDOR : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_CNDM) asm volatile(BODY_CNDM : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "vcc");
            if (KIND == K_CMP) asm volatile(BODY_CMP : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "vcc");
            if (KIND == K_PERM) asm volatile(BODY_PERM : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_BFE) asm volatile(BODY_BFE : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_LSHL64) asm volatile(BODY_LSHL64 : "+v"(w0), "+v"(w1));
            if (KIND == K_MULLO) asm volatile(BODY_MULLO : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_MAD24) asm volatile(BODY_MAD24 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_DPP) asm volatile(BODY_DPP : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_DPPBC) asm volatile(BODY_DPPBC : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_MBCNT) asm volatile(BODY_MBCNT : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_BCNT) asm volatile(BODY_BCNT : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_PKADD) asm volatile(BODY_PKADD : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_VAND) asm volatile(BODY_VAND : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_VSHL) asm volatile(BODY_VSHL : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_VSUB) asm volatile(BODY_VSUB : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_VMOV) asm volatile(BODY_VMOV : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_VADDLIT) asm volatile(BODY_VADDLIT : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_VADDK) asm volatile(BODY_VADDK : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_VADDCO) asm volatile(BODY_VADDCO : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "vcc");
            if (KIND == K_LSHLADD) asm volatile(BODY_LSHLADD : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_ADD3) asm volatile(BODY_ADD3 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_CNDS) asm volatile(BODY_CNDS : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "s20", "s21");
            if (KIND == K_CMPCND) asm volatile(BODY_CMPCND : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "vcc");
            if (KIND == K_CMPS) asm volatile(BODY_CMPS : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "s20", "s21", "s22", "s23");
            if (KIND == K_MOVDPP) asm volatile(BODY_MOVDPP : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_SDWA) asm volatile(BODY_SDWA : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_RFL) asm volatile(BODY_RFL : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
            if (KIND == K_DSREADU8)
                asm volatile("ds_read_u8 %0, %4\n ds_read_u8 %1, %4 offset:256\n ds_read_u8 %2, %4 offset:512\n ds_read_u8 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(la));
            if (KIND == K_DSWRITEB8)
                asm volatile("ds_write_b8 %0, %1\n ds_write_b8 %0, %2 offset:256\n ds_write_b8 %0, %3 offset:512\n ds_write_b8 %0, %4 offset:768"
                             : : "v"(la), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory");
            if (KIND == K_DSREAD64)
                asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:1024\n ds_read_b64 %0, %2 offset:2048\n ds_read_b64 %1, %2 offset:3072\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(w0), "=&v"(w1) : "v"(la16 >> 1));
            if (KIND == K_DSOR_RAND)
                asm volatile("ds_or_b32 %0, %1\n ds_or_b32 %0, %2 offset:4\n ds_or_b32 %0, %3 offset:8\n ds_or_b32 %0, %4 offset:12"
                             : : "v"(lrand), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory");
            if (KIND == K_DSREAD_RAND)
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %4 offset:8\n ds_read_b32 %3, %4 offset:12\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(lrand));
            if (KIND == K_SADD) asm volatile(BODY_SADD : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
            if (KIND == K_READLANE)
                asm volatile("v_readlane_b32 %0, %4, %1\n v_readlane_b32 %1, %5, %2\n v_readlane_b32 %2, %6, %3\n v_readlane_b32 %3, %7, %0\n"
                             "s_and_b32 %0, %0, 63\n s_and_b32 %1, %1, 63\n s_and_b32 %2, %2, 63\n s_and_b32 %3, %3, 63"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "scc");
            if (KIND == K_BPERM)
                asm volatile("ds_bpermute_b32 %0, %1, %0\n ds_bpermute_b32 %1, %2, %1\n ds_bpermute_b32 %2, %3, %2\n ds_bpermute_b32 %3, %0, %3\n s_waitcnt lgkmcnt(0)"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (KIND == K_DSREAD)
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(la));
            if (KIND == K_DSREAD128) {
                typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                v4 a, b, c, d;
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(la16));
                v0 ^= a.x ^ b.y ^ c.z ^ d.w;
            }
            if (KIND == K_DSOR)
                asm volatile("ds_or_b32 %0, %1\n ds_or_b32 %0, %2 offset:256\n ds_or_b32 %0, %3 offset:512\n ds_or_b32 %0, %4 offset:768"
                             : : "v"(la), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory");
            if (KIND == K_DSWRITE)
                asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:256\n ds_write_b32 %0, %3 offset:512\n ds_write_b32 %0, %4 offset:768"
                             : : "v"(la), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory");
            if (KIND == K_VALU_SALU)
                asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 %2, %2, %3\n v_add_u32 %1, %1, %0\n s_xor_b32 %3, %3, %2"
                             : "+v"(v0), "+v"(v1), "+s"(s0), "+s"(s1) : : "scc");
            if (KIND == K_BALLOT)
                asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_and_b64 %2, vcc, exec\n v_cmp_lt_u32 vcc, %1, %0\n s_and_b64 %3, vcc, exec"
                             : "+v"(v0), "+v"(v1), "=s"(w0), "=s"(w1) : : "vcc", "scc");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3 + (uint32_t)w0 + (uint32_t)w1 + lds[threadIdx.x];
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run_kind(uint32_t* d, uint64_t* dclk) {
    const int iters = 400;  // x 64 x 4 = 102 400 instructions per wave
    printf("%-32s", NAMES[KIND]);
    float ms5 = 0;
    for (int wps : {1, 2, 5, 8}) {
        const int grid = 256 * 4 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d, dclk, 4, 1u);
        hipDeviceSynchronize();
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d, dclk, iters, 1u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<uint64_t> c(grid);
        hipMemcpy(c.data(), dclk, grid * 8, hipMemcpyDeviceToHost);
        std::sort(c.begin(), c.end());
        const double med = (double)c[grid / 2];
        const double n_wave = iters * 64.0 * 4.0 * ((KIND == K_READLANE) ? 2.0 : 1.0);
        // clocks per instruction per SIMD: the median wave's clocks / (instructions of the wps waves sharing its SIMD)
        printf("  w%d: %6.2f clk (%.3f ms)", wps, med / (n_wave * wps), ms);
        if (wps == 5) ms5 = ms;
        if (wps == 8) printf("  | marginal %.2f ns per wave-instr per SIMD (w5->w8) = %.2f clk @2.4GHz", (ms - ms5) * 1e6 / (3.0 * n_wave), (ms - ms5) * 1e6 / (3.0 * n_wave) * 2.4);
        hipEventDestroy(a); hipEventDestroy(b);
    }
    printf("\n");
}

template <int K0>
static void run_all(uint32_t* d, uint64_t* dclk) {
    if constexpr (K0 < K_N) {
        run_kind<K0>(d, dclk);
        run_all<K0 + 1>(d, dclk);
    }
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    uint64_t* dclk; hipMalloc(&dclk, 256 * 4 * 8 * 8);
    printf("issue_test: clocks (s_memtime shader clocks) per wave64 instruction per SIMD, by waves per SIMD (w1/w2/w5/w8);\n"
           "grid = 256 CUs x 4 SIMDs x w workgroups of one wave. readlane rows count the paired s_and as an instruction too.\n");
    run_all<0>(d, dclk);
    return 0;
}
