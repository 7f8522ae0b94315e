// What does a 16-byte LDS gather cost on gfx950 when every lane reads its own (byte-unaligned) position of a 4 KiB window?
// (round 6: the output-owner executor reads match sources and literal runs with ds_read_b128 at arbitrary byte addresses)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_gather tools/src/lds_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T ld_u(const uint8_t* p) { T v; __builtin_memcpy(&v, p, sizeof(T)); return v; }

// mode 0: lane-linear (16 B per lane) + a; 1: random 16-aligned + a; 2: random byte address; 3: runs (lanes in groups of 4 read neighbouring 16 B, group base random + a)
template <typename T, int N>
__global__ void __launch_bounds__(64) thr(uint32_t* out, int iters, int mode, int a, uint32_t* check) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096 + 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096 + 64; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    uint32_t h = (lane * 2654435761u) ^ 0x9E3779B9u;
    uint32_t acc = 0, bad = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            h = h * 1664525u + 1013904223u;
            uint32_t addr;
            if (mode == 0) addr = (lane * 16 + j * 256 + a) & 4095;
            else if (mode == 1) addr = (((h >> 8) & 255u) * 16 + a) & 4095;
            else if (mode == 2) addr = (h >> 8) & 4095;
            else addr = ((((h >> 8) & 255u) >> 2 << 2) * 16 + (lane & 3) * 16 + a) & 4095;
            if (mode == 3) { uint32_t g = __shfl(h, lane & ~3); addr = ((((g >> 8) & 63u) * 64) + (lane & 3) * 16 + a) & 4095; }
            uint32_t w = 0;
#pragma unroll
            for (int k = 0; k < N; k++) {
                T r = ld_u<T>(lds + addr + k * sizeof(T));
                uint32_t t; __builtin_memcpy(&t, &r, 4);
                w += t;
                if (check && k == 0 && i == 0) {  // correctness of the first dword
                    const uint32_t b0 = (uint8_t)(addr * 7 + 3), b1 = (uint8_t)((addr + 1) * 7 + 3), b2 = (uint8_t)((addr + 2) * 7 + 3), b3 = (uint8_t)((addr + 3) * 7 + 3);
                    if (t != (b0 | b1 << 8 | b2 << 16 | b3 << 24)) bad++;
                    if (sizeof(T) == 16) { v4u q; __builtin_memcpy(&q, &r, 16); const uint32_t e = addr + 12;
                        const uint32_t c0 = (uint8_t)(e * 7 + 3), c1 = (uint8_t)((e + 1) * 7 + 3), c2 = (uint8_t)((e + 2) * 7 + 3), c3 = (uint8_t)((e + 3) * 7 + 3);
                        if (q.w != (c0 | c1 << 8 | c2 << 16 | c3 << 24)) bad++; }
                }
            }
            acc += w;
        }
        asm volatile("" ::: "memory");
    }
    if (acc == 0x12345) out[0] = acc;
    if (check && bad) atomicAdd(check, bad);
}

template <typename T, int N>
static void bench(const char* name, uint32_t* d, int mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-44s", name);
    for (int a : {0, 1, 4, 8}) {
        if (mode == 2 && a) continue;
        const int iters = 4000, wg = 256 * 8;  // 8 waves per CU
        hipMemset(d + 1, 0, 4);
        hipLaunchKernelGGL((thr<T, N>), dim3(wg), dim3(64), 0, 0, d, 2, mode, a, d + 1);
        hipDeviceSynchronize();
        uint32_t bad = 0; hipMemcpy(&bad, d + 1, 4, hipMemcpyDeviceToHost);
        hipEventRecord(e0);
        hipLaunchKernelGGL((thr<T, N>), dim3(wg), dim3(64), 0, 0, d, iters, mode, a, (uint32_t*)nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double per_cu = 8.0 * iters * 8;  // 16-byte gathers per CU
        printf("  a=%d: %6.1f clk%s", a, ms * 1e-3 * 2.4e9 / per_cu, bad ? " WRONG" : "");
    }
    printf("\n");
}

int main() {
    uint32_t* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    printf("clocks (2.4 GHz) per 64-lane 16-byte LDS gather per CU, 8 waves per CU; includes ~6 VALU of address arithmetic per gather\n");
    bench<v4u, 1>("ds_read_b128, lane-linear", d, 0);
    bench<v4u, 1>("ds_read_b128, random 16-byte slots", d, 1);
    bench<v4u, 1>("ds_read_b128, random byte address", d, 2);
    bench<v4u, 1>("ds_read_b128, runs of 4 lanes", d, 3);
    bench<v2u, 2>("2 x ds_read_b64, lane-linear", d, 0);
    bench<v2u, 2>("2 x ds_read_b64, random 16-byte slots", d, 1);
    bench<v2u, 2>("2 x ds_read_b64, random byte address", d, 2);
    bench<uint32_t, 4>("4 x ds_read_b32, lane-linear", d, 0);
    bench<uint32_t, 4>("4 x ds_read_b32, random 16-byte slots", d, 1);
    bench<uint32_t, 4>("4 x ds_read_b32, random byte address", d, 2);
    bench<uint32_t, 4>("4 x ds_read_b32, runs of 4 lanes", d, 3);
    return 0;
}
