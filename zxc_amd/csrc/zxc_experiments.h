// zxc_experiments.h — the ONE gate in front of every timing-ablation / A-B switch of the device sources.
//
// The kernels carry switches that exist for measurements only (tools/build_variant.sh, tools/build_enc_variant.sh): several of
// them produce WRONG OUTPUT on purpose (EXP_NO_FAR, ABL_*, EXP_ENC_NOWALK ...: timing only), the others change tuning constants.
// The product Makefile defines none of them and never defines ZXC_EXPERIMENT: a stray -D of any name below in a build without
// -DZXC_EXPERIMENT stops the compilation here instead of producing a silently different library. Included first by every
// device translation unit (before the constants' defaults: "defined" then means "given on the command line").
#ifndef ZXC_EXPERIMENTS_H
#define ZXC_EXPERIMENTS_H
#ifndef ZXC_EXPERIMENT
#if defined(ABL_ALL_NEAR) || defined(ABL_NO_DEPS) || defined(ABL_NO_FARPUT) || defined(ABL_NO_FARTAIL) || defined(ABL_NO_FLUSH) || \
    defined(ABL_NO_LIT) || defined(ABL_NO_LITTAIL) || defined(ABL_NO_NEAR) || defined(ASM_MARKERS) || defined(ENC_ALIGNED_CANDIDATES) || \
    defined(ENC_L3_HB) || defined(ENC_L4_HB) || defined(ENC_L57_HB) || defined(ENC_L57_NC) || defined(ENC_L67_CWB) || defined(ENC_U) || defined(EXP_ENC_CLOCKS) || \
    defined(EXP_ENC_ENV) || defined(EXP_ENC_EXTRA_SALU) || defined(EXP_ENC_EXTRA_VALU) || defined(EXP_ENC_NOPARSE) || defined(EXP_ENC_NOSTORE) || defined(EXP_ENC_NOWALK) || defined(EXP_ENC_WALK_ALL) || \
    defined(EXP_EXTRA_SLEEP) || defined(EXP_EXTRA_VMEM) || defined(EXP_EXTRA_SALU) || defined(EXP_EXTRA_VALU) || defined(EXP_NO_FAR) || defined(EXP_NO_OPTPARSE) || defined(EXP_NO_PIV_LDS) || \
    defined(EXP_NO_PRE) || defined(EXP_NT_FAR) || defined(EXP_NT_LIT) || defined(EXP_OPTPARSE_L7) || defined(EXP_PDIR_BADOUT) || \
    defined(EXP_PDIR_STOP1) || defined(EXP_PDIR_STOP2) || defined(EXP_PDIR_STOP3) || defined(EXP_PHASES) || defined(EXP_PIV_NOSTORE) || \
    defined(EXP_PIV_PROF) || defined(EXP_PRIO) || defined(EXP_RLE_FULL) || defined(EXP_SKIP_FULL) || defined(EXP_TIMES) || defined(FAR_EARLY) || \
    defined(FAR_PRELOAD) || defined(FLUSH_NT) || defined(LDS_FENCE_SCOPE) || defined(LEAN_OWNER) || defined(OWN_Q) || defined(OWN_C_STORE) || defined(LEAN_WAVES_PER_SIMD) || defined(LEAN_FLUSH_CHUNKS) || defined(LIT_LOOP2) || \
    defined(LIT_MED) || defined(LIT_PREFETCH) || defined(LIT_PREFETCH_AHEAD) || defined(LIT_PRELOAD) || defined(MATCH_MED) || \
    defined(PIV_FLAT_INFLIGHT) || defined(PIV_INFLIGHT) || defined(PIV_VARIANT_FILE) || defined(REDIRECT_PASSES) || defined(RING_BYTES) || \
    defined(SPARSE_MAX) || defined(TILE_MAX) || defined(WAVES_PER_SIMD) || defined(ZXC_RLE_LEAN_MAX_JOBS)
#error "an experiment / tuning switch is defined without -DZXC_EXPERIMENT: this is not the product configuration (zxc_experiments.h)"
#endif
#endif
#endif
