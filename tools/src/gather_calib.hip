// Micro-benchmark: what does rocprofv3's FETCH_SIZE report for the decode kernel's ACCESS PATTERNS?
// (the 1.98x factor was calibrated on a 16 B/lane streaming copy only.)
// Four read-only kernels over a 2 GiB buffer (>> the 256 MiB Infinity Cache), each issuing N wave-loads, non-temporal like
// the kernel's far reads:
//   stream16      lane i reads 16 B at 16 i (coalesced, 1 KiB per wave-load)             known bytes: 16 per lane-load
//   gather16_in64 every lane reads 16 B at a random 64-byte-aligned address + 16          one 64-byte sector per lane-load
//   gather16_any  every lane reads 16 B at a random BYTE address                         1 + 15/64 sectors per lane-load
//   gather32_any  every lane reads 2 x 16 B at a random byte address (a 32-byte match)   1 + 31/64 sectors per lane-load
// Run each under `rocprofv3 --pmc FETCH_SIZE` (and the TCC_EA0_RDREQ* pass): dispatches are named by kernel.
// Output: one line per kernel with the number of lane-loads and the ideal sector / line counts, to divide the counter by.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef v4u __attribute__((aligned(1))) v4u_unaligned;

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

extern "C" __global__ void __launch_bounds__(256) stream16(const uint8_t* __restrict__ buf, uint64_t bytes, uint32_t* out, int reps) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x, nthr = (uint64_t)gridDim.x * 256u;
    v4u acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; r++) {
        const uint64_t a = (tid + (uint64_t)r * nthr) * 16u;
        if (a + 16u <= bytes) acc ^= __builtin_nontemporal_load((const v4u*)(buf + a));
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

template <int MODE>  // 0: in one 64-byte sector, 1: any byte address, 2: 32 bytes at any byte address
__device__ __forceinline__ void gather_body(const uint8_t* __restrict__ buf, uint64_t bytes, uint32_t* out, int reps) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    v4u acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; r++) {
        const uint64_t h = mix(tid * 1315423911ull + (uint64_t)r * 0x9E3779B97F4A7C15ull + MODE);
        uint64_t a = h % (bytes - 256u);
        if (MODE == 0) a = (a & ~63ull) + 16u;
        acc ^= __builtin_nontemporal_load((const v4u_unaligned*)(buf + a));
        if (MODE == 2) acc ^= __builtin_nontemporal_load((const v4u_unaligned*)(buf + a + 16u));
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}
extern "C" __global__ void __launch_bounds__(256) gather16_in64(const uint8_t* b, uint64_t n, uint32_t* o, int reps) { gather_body<0>(b, n, o, reps); }
extern "C" __global__ void __launch_bounds__(256) gather16_any(const uint8_t* b, uint64_t n, uint32_t* o, int reps) { gather_body<1>(b, n, o, reps); }
extern "C" __global__ void __launch_bounds__(256) gather32_any(const uint8_t* b, uint64_t n, uint32_t* o, int reps) { gather_body<2>(b, n, o, reps); }

template <typename K>
static void time_it(const char* name, K kern, const uint8_t* buf, uint64_t bytes, uint32_t* out, int grid, int reps, double ideal_sectors_per_load,
                    double ideal_lines_per_load, double useful_bytes_per_load) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, bytes, out, reps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double loads = (double)grid * 256.0 * reps;
    printf("%-14s lane_loads %.0f  useful_bytes %.0f  ideal_64B_sector_bytes %.0f  ideal_128B_line_bytes %.0f  time_ms %.3f  useful_GBps %.1f  sector_GBps %.1f\n",
           name, loads, loads * useful_bytes_per_load, loads * ideal_sectors_per_load * 64.0, loads * ideal_lines_per_load * 128.0, ms,
           loads * useful_bytes_per_load / ms * 1e-6, loads * ideal_sectors_per_load * 64.0 / ms * 1e-6);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main(int argc, char** argv) {
    const uint64_t bytes = argc > 1 ? (uint64_t)atoll(argv[1]) << 20 : 2ull << 30;  // buffer size in MiB (default 2 GiB, far beyond the 256 MiB Infinity Cache)
    uint8_t* buf; uint32_t* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 32;  // 8 waves per SIMD
    // stream: 2 GiB = 134 217 728 lane-loads of 16 B = 16 reps of 8 388 608 threads... grid*256 = 2 097 152 threads -> 64 reps
    printf("buffer %llu MiB\n", (unsigned long long)(bytes >> 20));
    for (int rep = 0; rep < 2; rep++) {
    time_it("stream16", stream16, buf, bytes, out, grid, 64, 0.25, 0.125, 16.0);
    time_it("gather16_in64", gather16_in64, buf, bytes, out, grid, 16, 1.0, 1.0, 16.0);
    time_it("gather16_any", gather16_any, buf, bytes, out, grid, 16, 1.0 + 15.0 / 64.0, 1.0 + 15.0 / 128.0, 16.0);
    time_it("gather32_any", gather32_any, buf, bytes, out, grid, 16, 1.0 + 31.0 / 64.0, 1.0 + 31.0 / 128.0, 32.0);
    }
    return 0;
}
