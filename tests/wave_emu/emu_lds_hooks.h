// emu_lds_hooks.h — pre-included (-include) when the device sources are built for the CPU wave emulator:
// the LDS access points of zxc_amd/csrc/zxc_lds.h, routed through the emulator's model of a wave's LDS.
// Between two rendezvous of the wave a lane sees memory as it was at the last rendezvous plus its OWN
// writes; the writes of all lanes are applied at the next rendezvous. That is the lock-step semantics for
// code that follows the discipline stated in zxc_lds.h, and it makes code that breaks it fail:
// two lanes plain-storing the same byte, or a plain store meeting a ds_or of another lane, aborts.
#pragma once
#include <stdint.h>
#define ZXC_LDS_HOOKS 1
namespace emu {
void lds_write(void* p, const void* data, unsigned size, bool is_or);
void lds_read(const void* p, void* out, unsigned size);
void lds_read_unaligned(const void* p, void* out, unsigned size);
template <typename T> inline T lds_ld(const void* p) { T v; lds_read(p, &v, sizeof(T)); return v; }
template <typename T> inline void lds_st(void* p, T v) { lds_write(p, &v, sizeof(T), false); }
}
typedef uint32_t emu_v4u __attribute__((ext_vector_type(4)));
#define LDS_LD8(p) ((uint32_t)emu::lds_ld<uint8_t>((const void*)(p)))
#define LDS_LD32(p) (emu::lds_ld<uint32_t>((const void*)(p)))
#define LDS_LD128(p) (emu::lds_ld<emu_v4u>((const void*)(p)))
#define LDS_LD16(p) ((uint32_t)emu::lds_ld<uint16_t>((const void*)(p)))
#define LDS_LD64(p) (emu::lds_ld<uint64_t>((const void*)(p)))
#define LDS_ST16(p, v) emu::lds_st<uint16_t>((void*)(p), (uint16_t)(v))
#define LDS_ST64(p, v) emu::lds_st<uint64_t>((void*)(p), (uint64_t)(v))
#define LDS_LD128U(p) ([&] { emu_v4u v_; emu::lds_read_unaligned((const void*)(p), &v_, 16); return v_; }())
#define LDS_ST8(p, v) emu::lds_st<uint8_t>((void*)(p), (uint8_t)(v))
#define LDS_ST32(p, v) emu::lds_st<uint32_t>((void*)(p), (uint32_t)(v))
#define LDS_ST128(p, v) emu::lds_st<emu_v4u>((void*)(p), (v))
#define LDS_OR32(p, v) do { uint32_t v_ = (uint32_t)(v); emu::lds_write((void*)(p), &v_, 4, true); } while (0)
