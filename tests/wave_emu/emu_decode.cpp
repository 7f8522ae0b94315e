// emu_decode.cpp — the decode kernels of zxc_amd/csrc compiled for the CPU wave emulator and driven
// block by block (one emulated wavefront per block, like the real launch). Test infrastructure.
#include <functional>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "zxc_decode_kernel.hip"   // found via -I zxc_amd/csrc; <hip/hip_runtime.h> resolves to tests/wave_emu/hip/

namespace emu { void run_wave(const std::function<void()>& body, unsigned block, unsigned grid, int n_lanes); }
extern char __start_emu_lds[], __stop_emu_lds[];

static uint32_t emu_last_deferred = 0;
extern "C" __attribute__((visibility("default"))) uint32_t emu_last_deferred_count(void) { return emu_last_deferred; }

extern "C" __attribute__((visibility("default")))
int emu_decode_blocks(const uint8_t* comp, size_t comp_bytes, const zxc_dev_job_t* jobs, uint32_t n_jobs, uint8_t* out,
                      size_t out_bytes, int32_t* status, uint32_t block_size, int verify_trailer, const uint8_t* dict,
                      uint32_t dict_size, const uint8_t* dict_huf) {
    // padded private copies: the kernels read (never use) a few bytes past the ends, as they may in device buffers
    std::vector<uint8_t> c(comp_bytes + 8192, 0xEE), o(out_bytes + 8192, 0xDD);
    memcpy(c.data() + 4096, comp, comp_bytes);
    const uint32_t stride = (2u * (block_size + 64u) + block_size / 5u + 16u + 64u + 255u) & ~255u;
    const uint32_t n_slots = 4;
    std::vector<uint8_t> scratch((size_t)n_slots * stride + 4096, 0xCC);
    std::vector<uint32_t> busy(8192, 0);
    std::vector<uint8_t> dct;
    const uint8_t* dptr = nullptr;
    if (dict && dict_size) { dct.assign(dict_size + 8192, 0xBB); memcpy(dct.data() + 4096, dict, dict_size); dptr = dct.data() + 4096; }
    if (dptr || dict_huf) {
        for (uint32_t b = 0; b < n_jobs; b++) {
            memset(__start_emu_lds, 0xA5, (size_t)(__stop_emu_lds - __start_emu_lds));  // LDS is not zero at launch
            emu::run_wave([&] {
                zxc_decode_blocks_dict_kernel(c.data() + 4096, jobs, n_jobs, o.data() + 4096, status, block_size,
                                              verify_trailer ? 4u : 0u, scratch.data(), stride, 0u, busy.data(), n_slots,
                                              nullptr, 0u, dptr, dict_size, dict_huf);
            }, b, n_jobs, 64);
        }
    } else {
        // the two-pass launch of zxc_hip_shim.hip: the lean kernel over every block (in a launch order that is not the
        // identity), then the full kernel, a fixed grid walking the list of blocks the lean kernel handed over
        std::vector<uint32_t> order(n_jobs), list(n_jobs + 2u, 0u);
        for (uint32_t b = 0; b < n_jobs; b++) {
            order[b] = n_jobs - 1u - b;
            // (what zxc_order_scatter_kernel appends, by the same predicate)
            if (block_needs_full_kernel(c.data() + 4096 + jobs[order[b]].comp_off, jobs[order[b]].comp_size, verify_trailer ? 4u : 0u)) list[2u + list[0]++] = b;
        }
        for (uint32_t b = 0; b < n_jobs; b++) {
            memset(__start_emu_lds, 0xA5, (size_t)(__stop_emu_lds - __start_emu_lds));
            emu::run_wave([&] {
                zxc_decode_blocks_lean_kernel(c.data() + 4096, jobs, n_jobs, o.data() + 4096, status, block_size, order.data(), 0u, verify_trailer ? 4u : 0u);
            }, b, n_jobs, 64);
        }
        emu_last_deferred = list[0];
        const uint32_t grid = n_jobs < 3u ? n_jobs : 3u;
        for (uint32_t b = 0; b < grid; b++) {
            memset(__start_emu_lds, 0xA5, (size_t)(__stop_emu_lds - __start_emu_lds));
            emu::run_wave([&] {
                zxc_decode_blocks_kernel(c.data() + 4096, jobs, n_jobs, o.data() + 4096, status, block_size, verify_trailer ? 4u : 0u, scratch.data(),
                                         stride, 0u, busy.data(), n_slots, order.data(), 0u, list.data());
            }, b, grid, 64);
        }
    }
    memcpy(out, o.data() + 4096, out_bytes);
    return 0;
}
