// emu_decode.cpp — the decode kernels of zxc_amd/csrc compiled for the CPU wave emulator and driven
// block by block (one emulated wavefront per block, like the real launch). Test infrastructure.
#include <functional>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "zxc_decode_kernel.hip"   // found via -I zxc_amd/csrc; <hip/hip_runtime.h> resolves to tests/wave_emu/hip/

namespace emu { void run_wave(const std::function<void()>& body, unsigned block, unsigned grid, int n_lanes); }
extern char __start_emu_lds[], __stop_emu_lds[];

static uint32_t emu_last_deferred = 0, emu_last_pre = 0;
static size_t emu_pscratch_bytes = (size_t)8 << 20;  // scratch of the workgroup section decoder (0: every coded block goes to the full kernel)
extern "C" __attribute__((visibility("default"))) uint32_t emu_last_deferred_count(void) { return emu_last_deferred; }
extern "C" __attribute__((visibility("default"))) uint32_t emu_last_pre_count(void) { return emu_last_pre; }
extern "C" __attribute__((visibility("default"))) void emu_set_pscratch_bytes(size_t n) { emu_pscratch_bytes = n; }

extern "C" __attribute__((visibility("default")))
int emu_decode_blocks(const uint8_t* comp, size_t comp_bytes, const zxc_dev_job_t* jobs, uint32_t n_jobs, uint8_t* out,
                      size_t out_bytes, int32_t* status, uint32_t block_size, int verify_trailer, const uint8_t* dict,
                      uint32_t dict_size, const uint8_t* dict_huf) {
    // padded private copies: the kernels read (never use) a few bytes past the ends, as they may in device buffers
    std::vector<uint8_t> c(comp_bytes + 8192, 0xEE), o(out_bytes + 8192, 0xDD);
    memcpy(c.data() + 4096, comp, comp_bytes);
    const uint32_t stride = ZXC_DEV_SLOT_STRIDE(block_size);
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
        std::vector<uint32_t> order(n_jobs), list(n_jobs + 2u, 0u), plist(n_jobs + 4u, 0u);
        std::vector<zxc_dev_pre_t> pre(n_jobs);
        std::vector<uint8_t> pscratch(emu_pscratch_bytes + 4096, 0xC3);
        for (uint32_t b = 0; b < n_jobs; b++) {
            order[b] = n_jobs - 1u - b;
            // (what zxc_order_scatter_kernel records and appends, by the same predicate)
            const uint32_t i = order[b];
            uint32_t lit16, tok16, off = 0;
            uint32_t cls = classify_block(c.data() + 4096 + jobs[i].comp_off, jobs[i].comp_size, verify_trailer ? 4u : 0u, block_size,
                                          block_size + 2112u, lit16, tok16);
            if (cls == ZXC_DEV_CLS_PRE) {
                off = plist[2];
                plist[2] += lit16 + tok16;
                if ((uint64_t)off + lit16 + tok16 > (emu_pscratch_bytes >> 4)) cls = ZXC_DEV_CLS_FULL;
            }
            pre[i] = zxc_dev_pre_t{off, off + lit16, 0, cls};
            if (cls == ZXC_DEV_CLS_FULL) list[2u + list[0]++] = b;
            else if (cls == ZXC_DEV_CLS_PRE) plist[4u + plist[0]++] = i;
        }
        emu_last_pre = plist[0];
        if (plist[0]) {  // the workgroup section decoder: 512 threads = 8 emulated wavefronts sharing LDS
            const uint32_t wgs = plist[0] < 2u ? plist[0] : 2u;
            for (uint32_t g = 0; g < wgs; g++) {
                memset(__start_emu_lds, 0xA5, (size_t)(__stop_emu_lds - __start_emu_lds));
                emu::run_wave([&] { zxc_pivco_sections_kernel(c.data() + 4096, jobs, pre.data(), plist.data(), pscratch.data()); }, g, wgs, 512);
            }
        }
        for (uint32_t b = 0; b < n_jobs; b++) {
            memset(__start_emu_lds, 0xA5, (size_t)(__stop_emu_lds - __start_emu_lds));
            emu::run_wave([&] {
                zxc_decode_blocks_lean_kernel(c.data() + 4096, jobs, n_jobs, o.data() + 4096, status, block_size, order.data(), 0u, verify_trailer ? 4u : 0u,
                                              pre.data(), pscratch.data());
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
