// emu_encode.cpp — the encode kernels of zxc_amd/csrc compiled for the CPU wave emulator (one emulated
// wavefront per block, like the real launch). Test infrastructure: parity and compression-ratio checks of the
// match finder without a GPU.
#include <functional>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include <hip/hip_runtime.h>
#include <stdint.h>
namespace enc_tu {  // (zxc_rapidhash.inc defines non-inline device functions: keep this copy apart from emu_decode.cpp's)
#include "zxc_encode_kernel.hip"
}
using namespace enc_tu;

namespace emu { void run_wave(const std::function<void()>& body, unsigned block, unsigned grid, int n_lanes); }
extern char __start_emu_lds[], __stop_emu_lds[];

// level -> kernel entry + search effort, exactly as zxc_mi355x_encode_blocks_device (zxc_hip_shim.hip) picks them
#include "zxc_encode_levels.h"
static void zxc_encode_dispatch(int level, const uint8_t* src, uint64_t src_size, uint32_t block_size, uint8_t* slots,
                                uint32_t stride, uint32_t* sizes, uint32_t nb, uint32_t ck, uint32_t dict_size, uint8_t* hs) {
    zxc_enc_level_t p = zxc_enc_level_bs(level, block_size);
    if (const char* e = getenv("ZXC_EMU_ENC_ENTRY")) p.entry = atoi(e);  // (design experiments: another table geometry at the level's search effort)
    switch (p.entry) {
        case 0: zxc_encode_blocks_kernel_l1(src, src_size, block_size, slots, stride, sizes, nb, ck, p.depth, p.sufficient, p.lazy, dict_size, hs, p.huf); break;
        case 1: zxc_encode_blocks_kernel_l2(src, src_size, block_size, slots, stride, sizes, nb, ck, p.depth, p.sufficient, p.lazy, dict_size, hs, p.huf); break;
        case 2: zxc_encode_blocks_kernel_l3(src, src_size, block_size, slots, stride, sizes, nb, ck, p.depth, p.sufficient, p.lazy, dict_size, hs, p.huf); break;
        case 3: zxc_encode_blocks_kernel_l4(src, src_size, block_size, slots, stride, sizes, nb, ck, p.depth, p.sufficient, p.lazy, dict_size, hs, p.huf); break;
        case 4: zxc_encode_blocks_kernel_l57(src, src_size, block_size, slots, stride, sizes, nb, ck, p.depth, p.sufficient, p.lazy, dict_size, hs, p.huf); break;
        default: zxc_encode_blocks_kernel_l67(src, src_size, block_size, slots, stride, sizes, nb, ck, p.depth, p.sufficient, p.lazy, dict_size, hs, p.huf); break;
    }
}

extern "C" __attribute__((visibility("default")))
uint32_t emu_encode_slot_stride(uint32_t block_size) { return 2u * block_size + 512u; }

// slots: n_blocks * stride bytes, sizes: n_blocks entries (same contract as zxc_mi355x_encode_blocks_device)
extern "C" __attribute__((visibility("default")))
int emu_encode_blocks(const uint8_t* src, uint64_t src_size, uint32_t block_size, int level, int with_checksum,
                      uint8_t* slots, uint32_t* sizes, const uint8_t* dict, uint32_t dict_size) {
    const uint32_t nb = (uint32_t)((src_size + block_size - 1) / block_size);
    const uint32_t stride = emu_encode_slot_stride(block_size);
    std::vector<uint8_t> s(src_size + 8192, 0xEE), work;
    memcpy(s.data() + 4096, src, src_size);
    const uint8_t* in = s.data() + 4096;
    if (dict_size) {  // [dict | block] images, as zxc_mi355x_encode_blocks_dict_device prepares them
        work.assign((size_t)nb * ((size_t)block_size + dict_size) + 8192, 0xEE);
        for (uint32_t b = 0; b < nb; b++)
            emu::run_wave([&] { zxc_prepend_dict_kernel(s.data() + 4096, src_size, block_size, dict, dict_size, work.data() + 4096, nb); }, b, nb, 64);
        in = work.data() + 4096;
    }
    std::vector<uint8_t> hs((size_t)nb * 4u * ((size_t)block_size + 64u) + 8192, 0xEE);  // PivCo encoder scratch (levels 6-7)
    for (uint32_t b = 0; b < nb; b++) {
        memset(__start_emu_lds, 0xA5, (size_t)(__stop_emu_lds - __start_emu_lds));
        emu::run_wave([&] {
            zxc_encode_dispatch(level, in, src_size, block_size, slots, stride, sizes, nb, with_checksum ? 1u : 0u, dict_size, hs.data() + 4096);
        }, b, nb, 64);
    }
    return 0;
}
