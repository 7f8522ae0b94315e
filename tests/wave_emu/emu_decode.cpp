// emu_decode.cpp — the decode kernels of zxc_amd/csrc compiled for the CPU wave emulator and driven
// block by block (one emulated wavefront per block, like the real launch). Test infrastructure.
#include <functional>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "zxc_decode_kernel.hip"   // found via -I zxc_amd/csrc; <hip/hip_runtime.h> resolves to tests/wave_emu/hip/

namespace emu { void run_wave(const std::function<void()>& body, unsigned block, unsigned grid, int n_lanes); }
extern char __start_emu_lds[], __stop_emu_lds[];

static uint32_t emu_last_deferred = 0, emu_last_pre = 0, emu_last_secs[3] = {0, 0, 0};
static size_t emu_pscratch_bytes = (size_t)8 << 20;  // scratch of the workgroup section decoder (0: every coded block goes to the full kernel)
extern "C" __attribute__((visibility("default"))) uint32_t emu_last_deferred_count(void) { return emu_last_deferred; }
extern "C" __attribute__((visibility("default"))) uint32_t emu_last_pre_count(void) { return emu_last_pre; }
extern "C" __attribute__((visibility("default"))) void emu_set_pscratch_bytes(size_t n) { emu_pscratch_bytes = n; }
static size_t emu_rscratch_bytes = (size_t)4 << 20;  // scratch for the expanded literals of LEAN_RLE blocks (0: they go to the full kernel)
extern "C" __attribute__((visibility("default"))) void emu_set_rscratch_bytes(size_t n) { emu_rscratch_bytes = n; }
extern "C" __attribute__((visibility("default"))) uint32_t emu_last_section_count(int size_class) { return emu_last_secs[size_class]; }

// per-block checksums: 1 = hashed by zxc_block_checksum_kernel beside the decode and merged into the statuses (the product's plan without
// PRE blocks), 0 = inside the decode kernels (its plan with PRE blocks, dictionaries, the strict capacity)
static int emu_ck_apart = 1;
extern "C" __attribute__((visibility("default"))) void emu_set_ck_apart(int on) { emu_ck_apart = on; }

// the strict per-block capacity of zxc_decompress_block_safe (the kernels' cap_override argument; 0 = block_size + 2112)
static uint32_t emu_cap_override = 0;
extern "C" __attribute__((visibility("default"))) void emu_set_cap_override(uint32_t cap) { emu_cap_override = cap; }

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
    if (emu_cap_override && !(dptr || dict_huf)) {  // strict capacity: the full kernel alone, one block per workgroup (zxc_hip_shim.hip)
        for (uint32_t b = 0; b < n_jobs; b++) {
            memset(__start_emu_lds, 0xA5, (size_t)(__stop_emu_lds - __start_emu_lds));
            emu::run_wave([&] {
                zxc_decode_blocks_kernel(c.data() + 4096, jobs, n_jobs, o.data() + 4096, status, block_size, verify_trailer ? 4u : 0u, scratch.data(),
                                         stride, 0u, busy.data(), n_slots, nullptr, emu_cap_override, nullptr);
            }, b, n_jobs, 64);
        }
    } else if (dptr || dict_huf) {
        for (uint32_t b = 0; b < n_jobs; b++) {
            memset(__start_emu_lds, 0xA5, (size_t)(__stop_emu_lds - __start_emu_lds));  // LDS is not zero at launch
            emu::run_wave([&] {
                zxc_decode_blocks_dict_kernel(c.data() + 4096, jobs, n_jobs, o.data() + 4096, status, block_size,
                                              verify_trailer ? 4u : 0u, scratch.data(), stride, 0u, busy.data(), n_slots,
                                              nullptr, emu_cap_override, dptr, dict_size, dict_huf);
            }, b, n_jobs, 64);
        }
    } else {
        // the two-pass launch of zxc_hip_shim.hip, kernel by kernel: launch-order pass (histogram + scatter: order[], classes, work
        // lists), the section kernels of the three size classes, the lean kernel over every block, its second entry over the PRE
        // blocks, the full kernel over its list
        const uint32_t tb0 = verify_trailer ? 4u : 0u, g256 = (n_jobs + 255u) / 256u;
        const bool ck_apart = tb0 && emu_ck_apart;
        const uint32_t tb = ck_apart ? tb0 | ZXC_DEV_TRAILER_ELSEWHERE : tb0;
        std::vector<uint8_t> ck_bad(n_jobs + 16u, 0xEE);
        std::vector<uint32_t> hist(128, 0u), order(n_jobs), list(n_jobs + 2u, 0u), ctl(ZXC_DEV_CTL_WORDS, 0u), pre_entries(n_jobs);
        std::vector<zxc_dev_pre_t> pre(n_jobs);
        std::vector<zxc_dev_sec_t> secs(6u * (size_t)n_jobs);
        std::vector<uint8_t> pscratch(emu_pscratch_bytes + 4096, 0xC3);
        std::vector<uint8_t> rscratch(emu_rscratch_bytes + 4096, 0xC7);  // expanded literals of the LEAN_RLE blocks
        auto launch = [&](unsigned grid, int threads, const std::function<void()>& k) {
            for (unsigned g = 0; g < grid; g++) {
                memset(__start_emu_lds, 0xA5, (size_t)(__stop_emu_lds - __start_emu_lds));  // LDS is not zero at launch
                emu::run_wave(k, g, grid, threads);
            }
        };
        launch(g256, 256, [&] { zxc_order_hist_kernel(c.data() + 4096, jobs, n_jobs, block_size, hist.data()); });
        launch(g256, 256, [&] {
            zxc_order_scatter_kernel(c.data() + 4096, jobs, n_jobs, block_size, hist.data(), order.data(), list.data(), tb0, pre.data(), ctl.data(),
                                     pre_entries.data(), secs.data(), (uint32_t)(emu_pscratch_bytes >> 4), emu_cap_override ? emu_cap_override : block_size + 2112u,
                                     (uint32_t)(emu_rscratch_bytes >> 4));
        });
        emu_last_pre = ctl[ZXC_DEV_CTL_PRE];
        uint32_t* sec_hdr = ctl.data() + ZXC_DEV_CTL_SEC;
        for (int k = 0; k < 3; k++) emu_last_secs[k] = sec_hdr[2 * k];
        if (sec_hdr[0]) launch(2, 128, [&] { zxc_pivco_sections_small_kernel(c.data() + 4096, secs.data(), sec_hdr, pre.data(), pscratch.data()); });
        if (sec_hdr[2]) launch(2, 256, [&] { zxc_pivco_sections_medium_kernel(c.data() + 4096, secs.data() + 2u * (size_t)n_jobs, sec_hdr + 2, pre.data(), pscratch.data()); });
        if (sec_hdr[4]) launch(2, 512, [&] { zxc_pivco_sections_large_kernel(c.data() + 4096, secs.data() + 4u * (size_t)n_jobs, sec_hdr + 4, pre.data(), pscratch.data()); });
        if (ctl[ZXC_DEV_CTL_RLE_LIST]) {
            launch(3u, 64, [&] {
                zxc_rle_expand_kernel(c.data() + 4096, jobs, pre.data(), rscratch.data(), ctl.data() + ZXC_DEV_CTL_RLE_LIST, pre_entries.data() + n_jobs - 1u);
            });
        }
        launch(n_jobs, 64, [&] {
            zxc_decode_blocks_lean_kernel(c.data() + 4096, jobs, n_jobs, o.data() + 4096, status, block_size, order.data(), emu_cap_override, tb, pre.data(), rscratch.data());
        });
        if (ctl[ZXC_DEV_CTL_PRE]) launch(n_jobs, 64, [&] {
            zxc_decode_blocks_lean_pre_kernel(c.data() + 4096, jobs, o.data() + 4096, status, block_size, emu_cap_override, tb, pre.data(), pscratch.data(),
                                              ctl.data() + ZXC_DEV_CTL_PRE, pre_entries.data());
        });
        emu_last_deferred = list[0];
        const uint32_t grid = n_jobs < 3u ? n_jobs : 3u;
        for (uint32_t b = 0; b < grid; b++) {
            memset(__start_emu_lds, 0xA5, (size_t)(__stop_emu_lds - __start_emu_lds));
            emu::run_wave([&] {
                zxc_decode_blocks_kernel(c.data() + 4096, jobs, n_jobs, o.data() + 4096, status, block_size, tb, scratch.data(),
                                         stride, 0u, busy.data(), n_slots, order.data(), emu_cap_override, list.data());
            }, b, grid, 64);
        }
        if (ck_apart) {
            launch((n_jobs + 8u) / 9u, 64, [&] { zxc_block_checksum_kernel(c.data() + 4096, jobs, n_jobs, order.data(), ck_bad.data()); });
            launch(g256, 256, [&] { zxc_checksum_merge_kernel(ck_bad.data(), status, n_jobs); });
        }
    }
    memcpy(out, o.data() + 4096, out_bytes);
    return 0;
}
