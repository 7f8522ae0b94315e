// Micro-benchmark: does a second access to a cache line whose miss is still in flight ("hit on pending miss") cost the
// L1 more than an independent miss? 8 waves per SIMD gather 16 bytes per lane from random 128-byte lines of a 2 GiB buffer:
//   A  one x4 load per round (one miss per lane)
//   B  two x4 loads per round to two DIFFERENT random lines
//   C  x4 load + a dword load of the SAME line right behind it (the "fifth dword" pattern of the decode kernel)
//   D  x4 load, ~2000 clocks of ALU work, then the dword load of the same line (the line has arrived by then)
//   E  two x4 loads to the same line, back to back
// Reported: time per round and per lane-load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int MODE>
__global__ void __launch_bounds__(64) k(const uint8_t* __restrict__ buf, uint64_t bytes, uint32_t* out, int rounds) {
    const uint64_t tid = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    v4u acc = {0, 0, 0, 0};
    uint32_t acc1 = 0;
    for (int r = 0; r < rounds; r++) {
        const uint64_t h = mix(tid * 1315423911ull + (uint64_t)r * 0x9E3779B97F4A7C15ull);
        const uint8_t* a = buf + ((h % (bytes - 4096u)) & ~127ull);
        const uint8_t* b = buf + (((h >> 20) * 7919u % (bytes - 4096u)) & ~127ull);
        v4u v, w;
        uint32_t e = 0;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(a) : "memory");
        if (MODE == 1) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w) : "v"(b) : "memory"); }
        if (MODE == 2) { asm volatile("global_load_dword %0, %1, off offset:16" : "=v"(e) : "v"(a) : "memory"); }
        if (MODE == 3) {
            uint32_t x = (uint32_t)h;
#pragma unroll 1
            for (int i = 0; i < 250; i++) asm volatile("v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3" : "+v"(x));
            acc1 ^= x;
            asm volatile("global_load_dword %0, %1, off offset:16" : "=v"(e) : "v"(a) : "memory");
        }
        if (MODE == 4) { asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(w) : "v"(a) : "memory"); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc ^= v;
        if (MODE == 1 || MODE == 4) acc ^= w;
        acc1 ^= e;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w ^ acc1) == 0x12345u) out[0] = 1;
}
template <int MODE> static void run(const char* name, const uint8_t* buf, uint64_t bytes, uint32_t* out, int loads) {
    const int grid = 256 * 32, rounds = 200;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, buf, bytes, out, 4);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, buf, bytes, out, rounds);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-78s %.3f ms  %.1f ns per round per wave-slot  (%.2f G lane-lines/s)\n", name, ms, ms * 1e6 / rounds, (double)grid * 64 * rounds * (MODE == 1 ? 2 : 1) / ms * 1e-6);
}
int main() {
    const uint64_t bytes = 2ull << 30;
    uint8_t* buf; uint32_t* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64); hipMemset(buf, 1, bytes); hipDeviceSynchronize();
    run<0>("A one x4 gather per round", buf, bytes, out, 1);
    run<1>("B two x4 gathers, different lines", buf, bytes, out, 2);
    run<2>("C x4 gather + dword of the same line right behind", buf, bytes, out, 2);
    run<3>("D x4 gather, ~2000 clocks of ALU, dword of the same line", buf, bytes, out, 2);
    run<4>("E two x4 loads of the same line back to back", buf, bytes, out, 2);
    return 0;
}
