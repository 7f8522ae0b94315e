/* zxc_lds.h — the LDS access points of the sequence executor.
 *
 * Every LDS access of run_sequences() and its helpers goes through these, for two reasons:
 * the code says which DS instruction it means (aligned dword / 16-byte accesses, byte stores,
 * ds_or_b32), and a test harness may pre-define ZXC_LDS_HOOKS plus the same names to observe the
 * accesses (tests/wave_emu models the wave's lock-step LDS semantics on a CPU with them).
 * Discipline assumed by callers: a value written by one lane may be read by ANOTHER lane only
 * after wave_lds_fence(); write-after-read between lanes needs nothing (one wave's DS
 * instructions execute in program order). */
#ifndef ZXC_LDS_H
#define ZXC_LDS_H
#ifndef ZXC_LDS_HOOKS
#define LDS_LD8(p) ((uint32_t)*(const uint8_t*)(p))
#define LDS_LD32(p) (*(const uint32_t*)(p))
#define LDS_LD128(p) (*(const v4u*)(p))
#define LDS_ST8(p, v) (*(uint8_t*)(p) = (uint8_t)(v))
#define LDS_ST32(p, v) (*(uint32_t*)(p) = (uint32_t)(v))
#define LDS_ST128(p, v) (*(v4u*)(p) = (v))
/* round 6 (zxc_seq_own.inc): 16-bit / 64-bit accesses and a 16-byte read at ANY byte address (ds_read_b128 in the unaligned
 * access mode gfx950 runs in: correct at every alignment, ~65 clk per 64-lane gather per CU instead of ~45, profiles/r6a_lds_gather.log) */
#define LDS_LD16(p) ((uint32_t)*(const uint16_t*)(p))
#define LDS_ST16(p, v) (*(uint16_t*)(p) = (uint16_t)(v))
#define LDS_LD64(p) (*(const uint64_t*)(p))
#define LDS_ST64(p, v) (*(uint64_t*)(p) = (uint64_t)(v))
#define LDS_LD128U(p) (*(const v4u_unaligned*)(p))
/* ds_or_b32 (no return): bytes of different lanes meet in one dword without a read-modify-write in registers */
#define LDS_OR32(p, v) ((void)__hip_atomic_fetch_or((uint32_t*)(p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#endif
#endif
