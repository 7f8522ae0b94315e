// How many single-wave workgroups does one gfx950 CU really hold? Each workgroup spins for a fixed
// wall time; launch 256*k workgroups and see at which k the kernel time doubles.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/occ_test tools/src/occ_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(64) spin(uint32_t* out, uint64_t ticks) {
    extern __shared__ uint32_t dyn[];
    const uint64_t t0 = wall_clock64();
    uint32_t acc = 0;
    while (wall_clock64() - t0 < ticks) acc += dyn[threadIdx.x & 15];
    if (acc == 0x12345u) out[0] = acc;
}
int main() {
    uint32_t* d; hipMalloc(&d, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint64_t ticks = 100000;  // 1 ms at 100 MHz
    for (int lds : {0, 2048, 4096, 5392, 8144, 9216}) {
        hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        printf("LDS %5d B:", lds);
        for (int k : {8, 12, 16, 17, 18, 19, 20, 24, 28, 29, 30, 32, 33, 40}) {
            hipLaunchKernelGGL(spin, dim3(256 * k), dim3(64), lds, 0, d, 100);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(spin, dim3(256 * k), dim3(64), lds, 0, d, ticks);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
            printf(" k=%d:%.2f", k, ms);
        }
        printf("\n");
    }
    return 0;
}
