// Micro-benchmark: is v_cndmask_b32 (VOP2 form, implicit VCC) slow on gfx950 when VCC was last written by the scalar unit?
// issue_test measured 22 clk per wave-instruction for a stream of VOP2 v_cndmask reading a VCC nobody wrote, 4 clk for the VOP3 form
// with an SGPR-pair mask, ~3 clk when a VALU compare wrote VCC right before. This pins which producer / encoding is the slow one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ void __launch_bounds__(64) k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t v0 = threadIdx.x + seed, v1 = v0 * 3u + 1u, v2 = (v0 ^ 5u), v3 = (v0 + 7u);
    uint64_t m = 0x5555aaaa5555aaaaull ^ seed;
    if (KIND == 1) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(v0), "v"(v1) : "vcc");       // VALU wrote VCC once, before the loop
    if (KIND == 2) asm volatile("s_mov_b64 vcc, %0" : : "s"(m) : "vcc");                          // SALU wrote VCC once, before the loop
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++) {
            if (KIND <= 2)  // VOP2 form, VCC written before the loop (0: never)
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "vcc");
            if (KIND == 3)  // SALU writes VCC, then 4 VOP2 selects (what the compiler emits for a uniform ? : on vectors)
                asm volatile("s_mov_b64 vcc, %4\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(m) : "vcc");
            if (KIND == 4)  // VALU compare writes VCC, then 4 VOP2 selects
                asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "vcc");
            if (KIND == 5)  // VOP3 form with VCC as an explicit mask operand, SALU-written
                asm volatile("s_mov_b64 vcc, %4\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %3, %3, %0, vcc"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(m) : "vcc");
            if (KIND == 6)  // VOP3 form, SGPR pair written by SALU each time
                asm volatile("s_mov_b64 s[20:21], %4\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %0, s[20:21]"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(m) : "s20", "s21");
            if (KIND == 7)  // VOP2 v_and with an SGPR operand written by SALU right before (VALU reads a fresh SGPR)
                asm volatile("s_mov_b32 s20, %4\n v_and_b32 %0, s20, %0\n v_or_b32 %1, s20, %1\n v_and_b32 %2, s20, %2\n v_or_b32 %3, s20, %3"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"((uint32_t)m) : "s20");
            if (KIND == 8)  // v_cmp to SGPR pair, s_and with exec-like mask, VOP3 select (the ballot -> mask -> select chain)
                asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n s_and_b64 s[20:21], s[20:21], %4\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %0, s[20:21]"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(m) : "s20", "s21", "scc");
            if (KIND == 9)  // s_and_saveexec / s_or exec around one VALU instruction (an `if` on a divergent condition)
                asm volatile("s_and_saveexec_b64 s[20:21], %4\n v_add_u32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]\n v_add_u32 %1, %1, %2"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(m) : "s20", "s21", "scc");
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = v0 + v1 + v2 + v3;
}
template <int KIND>
static void run(const char* name, int per_body, uint32_t* d) {
    printf("%-64s", name);
    float ms5 = 0;
    for (int wps : {1, 5, 8}) {
        const int grid = 256 * 4 * wps, iters = 400;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d, 4, 1u);
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d, iters, 1u);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf(" w%d %.3f ms", wps, ms);
        if (wps == 5) ms5 = ms;
        if (wps == 8) printf(" | marginal %.2f clk @2.4GHz per instruction (%d per body)", (ms - ms5) * 1e6 / (3.0 * iters * 64.0 * per_body) * 2.4, per_body);
    }
    printf("\n");
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    run<0>("VOP2 cndmask, VCC never written", 4, d);
    run<1>("VOP2 cndmask, VCC written once by v_cmp before the loop", 4, d);
    run<2>("VOP2 cndmask, VCC written once by s_mov before the loop", 4, d);
    run<3>("s_mov vcc + 4 x VOP2 cndmask vcc", 5, d);
    run<4>("v_cmp vcc + 4 x VOP2 cndmask vcc", 5, d);
    run<5>("s_mov vcc + 4 x VOP3 cndmask vcc", 5, d);
    run<6>("s_mov s[20:21] + 4 x VOP3 cndmask s[20:21]", 5, d);
    run<7>("s_mov s20 + 4 x VOP2 and/or with s20", 5, d);
    run<8>("v_cmp sgpr + s_and + 2 x VOP3 cndmask", 4, d);
    run<9>("s_and_saveexec + v_add + s_or exec + v_add", 4, d);
    return 0;
}
