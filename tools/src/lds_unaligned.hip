// Are byte-granular (unaligned) LDS accesses of every width correct on gfx950, and what do they cost?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_unaligned tools/src/lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ T ld_u(const uint8_t* p) { T v; __builtin_memcpy(&v, p, sizeof(T)); return v; }
template <typename T> __device__ __forceinline__ void st_u(uint8_t* p, T v) { __builtin_memcpy(p, &v, sizeof(T)); }

// correctness: each lane writes W bytes at offset base(lane)+a, reads back at the same offset and around it
template <typename T>
__global__ void check(uint32_t* bad) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 64];
    const int lane = threadIdx.x;
    uint32_t errs = 0;
    for (int a = 0; a < 16; a++) {
        for (int i = lane; i < 64 * 64; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
        __syncthreads();
        uint8_t* p = lds + lane * 64 + 16 + a;
        T v;
        uint8_t pat[sizeof(T)];
        for (unsigned k = 0; k < sizeof(T); k++) pat[k] = (uint8_t)(0xA0 + k + lane);
        __builtin_memcpy(&v, pat, sizeof(T));
        st_u<T>(p, v);
        __syncthreads();
        // bytes around must be untouched, bytes inside must match
        for (int k = -4; k < (int)sizeof(T) + 4; k++) {
            const int idx = lane * 64 + 16 + a + k;
            const uint8_t want = (k >= 0 && k < (int)sizeof(T)) ? pat[k] : (uint8_t)(idx * 7 + 3);
            if (lds[idx] != want) errs++;
        }
        T r = ld_u<T>(p);
        uint8_t got[sizeof(T)];
        __builtin_memcpy(got, &r, sizeof(T));
        for (unsigned k = 0; k < sizeof(T); k++) if (got[k] != pat[k]) errs++;
        __syncthreads();
    }
    if (errs) atomicAdd(bad, errs);
}

// throughput: dependent-free stream of reads (or writes) at byte offset `a` per lane
template <typename T, bool WR>
__global__ void __launch_bounds__(64) thr(uint32_t* out, int iters, int a, int stride) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8192; i += 64) lds[i] = (uint8_t)i;
    __syncthreads();
    uint8_t* p = lds + ((lane * stride + a) & 4095);
    uint32_t acc = 0;
    T v; __builtin_memset(&v, 1, sizeof(T));
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (WR) { st_u<T>(p + j * 256, v); }
            else { T r = ld_u<T>(p + j * 256); uint32_t w; __builtin_memcpy(&w, &r, 4); acc += w; }
        }
        asm volatile("" ::: "memory");
    }
    if (acc == 0x12345) out[0] = acc;
}

template <typename T, bool WR>
static void bench(const char* name, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-14s", name);
    for (int a : {0, 1, 2, 4, 8}) {
        const int iters = 20000, wg = 256 * 8;  // 8 waves per CU
        hipLaunchKernelGGL((thr<T, WR>), dim3(wg), dim3(64), 0, 0, d, 10, a, 16);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((thr<T, WR>), dim3(wg), dim3(64), 0, 0, d, iters, a, 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double per_cu = 8.0 * iters * 8;  // wave-instructions per CU
        printf("  a=%d: %6.1f clk/instr/CU", a, ms * 1e-3 * 2.3e9 / per_cu);
    }
    printf("\n");
}

int main() {
    uint32_t* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    uint32_t h = 0;
    hipLaunchKernelGGL(check<uint16_t>, dim3(1), dim3(64), 0, 0, d); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("b16 errors %u\n", h); hipMemset(d, 0, 4);
    hipLaunchKernelGGL(check<uint32_t>, dim3(1), dim3(64), 0, 0, d); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("b32 errors %u\n", h); hipMemset(d, 0, 4);
    hipLaunchKernelGGL(check<v2u>, dim3(1), dim3(64), 0, 0, d); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("b64 errors %u\n", h); hipMemset(d, 0, 4);
    hipLaunchKernelGGL(check<v4u>, dim3(1), dim3(64), 0, 0, d); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("b128 errors %u\n", h); hipMemset(d, 0, 4);
    printf("throughput, lane stride 16 B, 8 waves/CU (clk assumes 2.3 GHz):\n");
    bench<uint32_t, false>("read b32", d); bench<v2u, false>("read b64", d); bench<v4u, false>("read b128", d);
    bench<uint32_t, true>("write b32", d); bench<v2u, true>("write b64", d); bench<v4u, true>("write b128", d);
    bench<uint16_t, true>("write b16", d); bench<uint8_t, true>("write b8", d);
    return 0;
}
