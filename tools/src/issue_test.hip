// Micro-benchmark: do VALU and SALU instructions of different waves on one SIMD co-issue on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/issue_test tools/src/issue_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>  // 0: VALU only, 1: SALU only, 2: VALU+SALU interleaved (same counts of each as 0 and 1), 3: VALU+LDS
__global__ void __launch_bounds__(64) k(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[1024];
    uint32_t v0 = threadIdx.x + seed, v1 = v0 * 3u, v2 = v0 ^ 5u, v3 = v0 + 7u;
    uint32_t s0 = seed, s1 = seed * 3u, s2 = seed ^ 9u, s3 = seed + 11u;
    lds[threadIdx.x] = v0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (MODE == 0 || MODE == 2 || MODE == 3) {
                asm volatile("v_xor_b32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_add_u32 %3, %3, %0"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            }
            if (MODE == 1 || MODE == 2) {
                asm volatile("s_xor_b32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_xor_b32 %2, %2, %3\n s_add_u32 %3, %3, %0"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
            }
            if (MODE == 3) {
                uint32_t t;
                asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %2, %2, %0" : "=&v"(t), "+v"(v0), "+v"(v1));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3;
}

template <int MODE>
static double run(int wg_per_cu, int iters, uint32_t* d) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * wg_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, 1u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 64 * 64 * 4);
    const int iters = 20000;  // x16x4 = 1.28M instructions of each kind per wave
    for (int w : {4, 8, 16, 32}) {  // workgroups (waves) per CU -> 1,2,4,8 per SIMD
        double v = run<0>(w, iters, d), s = run<1>(w, iters, d), vs = run<2>(w, iters, d), vl = run<3>(w, iters, d);
        double n = 1.28e6 * 16.0 / 16.0;
        printf("waves/SIMD %d: VALU %.2f ms  SALU %.2f ms  VALU+SALU %.2f ms  VALU+LDS(dep) %.2f ms | clk/instr/wave VALU %.2f (assuming 2.4GHz)\n",
               w / 4, v, s, vs, vl, v * 1e-3 * 2.4e9 / (iters * 64.0) / (w / 4.0));
        (void)n;
    }
    return 0;
}
