// Micro-benchmark: what does one wave64 vector-memory instruction cost in the CU's address / L1 pipeline, by address pattern?
// All data is L1/L2-resident (a 64 KiB buffer per workgroup slot, reused), so this is the issue / tag / data-return cost of the
// instruction itself, not memory bandwidth. One wave per workgroup; 1 / 4 / 8 waves per SIMD; the marginal cost per instruction
// per CU is taken from the 4 -> 8 step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef v4u __attribute__((aligned(1))) v4u_una;

enum { P_X4_CONTIG, P_X4_SAME, P_X4_LINE_EACH, P_X4_LINE_EACH_UNA, P_X4_8LANES, P_X4_8LANES_SAMEREST, P_X4_NEAR, P_X1_LINE_EACH, P_U8_CONTIG,
       P_X4_STORE, P_X4_16LANES, P_X4_PAIRS, P_X4_P4, P_X4_P8, P_X4_P1, P_X4_P2, P_X2_ALIGNED, P_X2_P4, P_X2_P1, P_X1_ALIGNED, P_X1_P1, P_X4_NT_P5, P_X4_P5_32LANES, P_X4_NEAR_ALIGNED, P_X4_STORE_ASM, P_X4_P16, P_N };
static const char* NAMES[P_N] = {
    "load x4, lane i at 16 i (1 KiB contiguous)", "load x4, every lane the same address", "load x4, every lane its own 128-byte line",
    "load x4, own line, unaligned (+5 bytes)", "load x4, 8 lanes active (exec), own lines", "load x4, 8 lanes own lines + 56 lanes one shared address",
    "load x4, lane i at 4 i + (i & 3) (overlapping, 260 bytes: the literal pattern)", "load dword, every lane its own line", "load ubyte, lane i at i (64 bytes)",
    "store x4 (volatile), lane i at 16 i", "load x4, 16 lanes active, own lines", "load x4, lanes in pairs on 32 lines",
    "load x4, own line +4 bytes (dword aligned)", "load x4, own line +8 bytes", "load x4, own line +1 byte", "load x4, own line +2 bytes",
    "load x2, own line, aligned", "load x2, own line +4", "load x2, own line +1", "load dword, own line, aligned", "load dword, own line +1",
    "load x4 nt, own line +5", "load x4, own line +5, 32 lanes active", "load x4, lane i at 4 i (dword aligned, overlapping 268 bytes)",
    "store x4 (plain), lane i at 16 i", "load x4, own line +16 (16-byte aligned, second sector quarter)"};

template <int P>
__global__ void __launch_bounds__(64) k(uint8_t* buf, uint32_t* out, int iters) {
    uint8_t* base = buf + (size_t)(blockIdx.x % 2048u) * 65536u;
    const uint32_t lane = threadIdx.x;
    uint32_t off;
    switch (P) {
        case P_X4_CONTIG: case P_X4_STORE: off = 16u * lane; break;
        case P_X4_SAME: off = 64u; break;
        case P_X4_LINE_EACH: case P_X1_LINE_EACH: case P_X4_8LANES: case P_X4_16LANES: off = 128u * lane; break;
        case P_X4_LINE_EACH_UNA: off = 128u * lane + 5u; break;
        case P_X4_8LANES_SAMEREST: off = (lane & 7u) == 0u ? 128u * lane : 0u; break;
        case P_X4_NEAR: off = 4u * lane + (lane & 3u); break;
        case P_U8_CONTIG: off = lane; break;
        case P_X4_PAIRS: off = 128u * (lane >> 1) + 16u * (lane & 1u); break;
        case P_X4_P4: case P_X2_P4: off = 128u * lane + 4u; break;
        case P_X4_P8: off = 128u * lane + 8u; break;
        case P_X4_P1: case P_X2_P1: case P_X1_P1: off = 128u * lane + 1u; break;
        case P_X4_P2: off = 128u * lane + 2u; break;
        case P_X2_ALIGNED: case P_X1_ALIGNED: off = 128u * lane; break;
        case P_X4_NT_P5: case P_X4_P5_32LANES: off = 128u * lane + 5u; break;
        case P_X4_NEAR_ALIGNED: off = 4u * lane; break;
        case P_X4_STORE_ASM: off = 16u * lane; break;
        case P_X4_P16: off = 128u * lane + 16u; break;
        default: off = 0;
    }
    v4u acc = {0, 0, 0, 0};
    uint32_t acc1 = 0;
    const bool active = (P == P_X4_P5_32LANES) ? ((lane & 1u) == 0u) : (P == P_X4_8LANES) ? ((lane & 7u) == 0u) : (P == P_X4_16LANES) ? ((lane & 3u) == 0u) : true;
    for (int i = 0; i < iters; i++) {
        const uint32_t o = off + ((uint32_t)i & 3u) * 8192u;  // four 8 KiB windows: stays inside the L1
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint8_t* a = base + o + (uint32_t)j * 0u;
            if (P == P_X1_LINE_EACH) { acc1 ^= *(volatile const uint32_t*)a; }
            else if (P == P_U8_CONTIG) { acc1 ^= *(volatile const uint8_t*)a; }
            else if (P == P_X4_STORE) { *(volatile v4u*)a = acc; }
            else if (P == P_X2_ALIGNED || P == P_X2_P4 || P == P_X2_P1) { uint64_t v; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(a) : "memory"); acc1 ^= (uint32_t)v; }
            else if (P == P_X1_ALIGNED || P == P_X1_P1) { uint32_t v; asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(a) : "memory"); acc1 ^= v; }
            else if (P == P_X4_NT_P5) { v4u v; asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(a) : "memory"); acc ^= v; }
            else if (P == P_X4_STORE_ASM) { asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(a), "v"(acc) : "memory"); }
            else if (active) { v4u v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(a) : "memory"); acc ^= v; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * 64 + lane] = acc.x ^ acc.y ^ acc.z ^ acc.w ^ acc1;
}

template <int P>
static void run(uint8_t* buf, uint32_t* out) {
    printf("%-82s", NAMES[P]);
    float ms4 = 0;
    const int iters = 2000;  // x 8 instructions
    for (int wps : {1, 4, 8}) {
        const int grid = 256 * 4 * wps;
        hipLaunchKernelGGL(k<P>, dim3(grid), dim3(64), 0, 0, buf, out, 16);
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<P>, dim3(grid), dim3(64), 0, 0, buf, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf(" w%d %.3f ms", wps, ms);
        if (wps == 4) ms4 = ms;
        if (wps == 8) printf(" | marginal %.1f clk @2.4GHz per instruction per CU", (ms - ms4) * 1e6 * 2.4 / (16.0 * iters * 8.0));
    }
    printf("\n");
}
template <int P0> static void run_all(uint8_t* b, uint32_t* o) { if constexpr (P0 < P_N) { run<P0>(b, o); run_all<P0 + 1>(b, o); } }
int main() {
    uint8_t* buf; uint32_t* out;
    hipMalloc(&buf, (size_t)2048 * 65536 + 65536); hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    hipMemset(buf, 1, (size_t)2048 * 65536 + 65536);
    run_all<0>(buf, out);
    return 0;
}
