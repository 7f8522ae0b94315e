/* zxc_error.h — error codes. Negative return = failure, same numbering as the
 * reference (include/zxc_error.h:38-74) so callers' switch statements keep working.
 * Two additional codes, outside the reference's range, report GPU conditions: this
 * library never silently falls back to a CPU decoder. */
#ifndef ZXC_ERROR_H
#define ZXC_ERROR_H
#include "zxc_export.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ZXC_OK = 0,
    ZXC_ERROR_MEMORY = -1,
    ZXC_ERROR_DST_TOO_SMALL = -2,
    ZXC_ERROR_SRC_TOO_SMALL = -3,
    ZXC_ERROR_BAD_MAGIC = -4,
    ZXC_ERROR_BAD_VERSION = -5,
    ZXC_ERROR_BAD_HEADER = -6,
    ZXC_ERROR_BAD_CHECKSUM = -7,
    ZXC_ERROR_CORRUPT_DATA = -8,
    ZXC_ERROR_BAD_OFFSET = -9,
    ZXC_ERROR_OVERFLOW = -10,
    ZXC_ERROR_IO = -11,
    ZXC_ERROR_NULL_INPUT = -12,
    ZXC_ERROR_BAD_BLOCK_TYPE = -13,
    ZXC_ERROR_BAD_BLOCK_SIZE = -14,
    ZXC_ERROR_DICT_REQUIRED = -15,
    ZXC_ERROR_DICT_MISMATCH = -16,
    ZXC_ERROR_DICT_TOO_LARGE = -17,
    ZXC_ERROR_BAD_LEVEL = -18,
    /* MI355X build only */
    ZXC_ERROR_GPU_UNAVAILABLE = -100, /* no HIP device, or a HIP runtime call failed */
    ZXC_ERROR_GPU_UNSUPPORTED = -101  /* valid input using a feature the device path lacks */
} zxc_error_t;

/* reference: zxc_error_name, include/zxc_error.h:84 */
ZXC_EXPORT const char* zxc_error_name(const int code);

#ifdef __cplusplus
}
#endif
#endif
