/* zxc_dev.h — types shared by the HIP kernels and the host-side C-ABI shim. */
#ifndef ZXC_DEV_H
#define ZXC_DEV_H
#include "../../include/zxc_mi355x.h" /* zxc_dev_job_t */

/* Device-side status that is not a reference zxc_error_t value: the block is valid
 * but uses a section coding this kernel cannot decode yet (maps to
 * ZXC_ERROR_GPU_UNSUPPORTED on the host; never a silent CPU fallback). */
#define ZXC_DEV_E_UNSUPPORTED (-101)
#define ZXC_DEV_DEFER (-103)      /* lean kernel only, never stored: the block goes to the full kernel's list */
#define ZXC_DEV_E_INTERNAL (-102) /* kernel self-check tripped (a bug, never an input property) */
#endif
