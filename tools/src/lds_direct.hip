// global_load_lds_dwordx4 on gfx950: where does lane l's data land (M0 base + 16 l?), any source alignment, and what does it cost per CU?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_direct tools/src/lds_direct.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define GLDS(gp, lp) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)

__global__ void check(const uint8_t* __restrict__ in, uint32_t* bad, int unal) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) stage[i] = 0xEE;
    __syncthreads();
    const uint32_t o0 = (lane * 977u) % 60000u + unal, o1 = (lane * 131u + 7u) % 60000u + 2 * unal;
    GLDS(in + o0, stage);
    GLDS(in + o1, stage + 1024);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    uint32_t errs = 0;
    for (int k = 0; k < 16; k++) {
        if (stage[16 * lane + k] != in[o0 + k]) errs++;
        if (stage[1024 + 16 * lane + k] != in[o1 + k]) errs++;
    }
    if (errs) atomicAdd(bad, errs);
}
// cost: N LDS-direct loads (gather, own line) vs N register loads per iteration
template <bool DIRECT>
__global__ void __launch_bounds__(64) thr(const uint8_t* __restrict__ in, uint32_t* out, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[4096];
    const int lane = threadIdx.x;
    uint32_t h = (blockIdx.x * 64 + lane) * 2654435761u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        const uint8_t* p = in + ((h >> 4) & 0xFFFFF0u) + 5;
        if (DIRECT) {
            GLDS(p, stage); GLDS(p + 4096, stage + 1024); GLDS(p + 8192, stage + 2048); GLDS(p + 12288, stage + 3072);
            __builtin_amdgcn_s_waitcnt(0);
            const v4u a = *(v4u*)(stage + 16 * lane), b = *(v4u*)(stage + 1024 + 16 * lane), c = *(v4u*)(stage + 2048 + 16 * lane), d = *(v4u*)(stage + 3072 + 16 * lane);
            acc += a.x + b.y + c.z + d.w;
        } else {
            v4u a, b, c, d;
            __builtin_memcpy(&a, p, 16); __builtin_memcpy(&b, p + 4096, 16); __builtin_memcpy(&c, p + 8192, 16); __builtin_memcpy(&d, p + 12288, 16);
            acc += a.x + b.y + c.z + d.w;
        }
    }
    if (acc == 0x12345) out[0] = acc;
}
int main() {
    const size_t N = 32u << 20;
    std::vector<uint8_t> h(N);
    for (size_t i = 0; i < N; i++) h[i] = (uint8_t)(i * 131 + (i >> 9));
    uint8_t* d; hipMalloc(&d, N + 65536); hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    uint32_t* bad; hipMalloc(&bad, 64); hipMemset(bad, 0, 64);
    for (int u = 0; u < 4; u++) {
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(check, dim3(8), dim3(64), 0, 0, d, bad, u);
        uint32_t b = 0; hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
        printf("global_load_lds_dwordx4, source alignment +%d (+%d): lane l's 16 bytes at base + 16 l: %s (%u byte errors)\n", u, 2 * u, b ? "NO" : "yes", b);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int direct = 0; direct < 2; direct++) {
        const int iters = 2000, wg = 256 * 8;
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (direct) hipLaunchKernelGGL((thr<true>), dim3(wg), dim3(64), 0, 0, d, bad + 4, iters);
            else hipLaunchKernelGGL((thr<false>), dim3(wg), dim3(64), 0, 0, d, bad + 4, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%s: 4 gathers of 16 B (own lines, +5 bytes) per iteration, 8 waves/CU: %.3f ms = %.0f clk per gather instruction per CU\n",
               direct ? "LDS-direct + ds_read_b128" : "register loads           ", ms, ms * 1e-3 * 2.4e9 / (8.0 * iters * 4));
    }
    return 0;
}
