/* zxc_constants.h — public constants of the ZXC API (values are the wire/ABI contract).
 * Replaces reference include/zxc_constants.h:20-134. */
#ifndef ZXC_CONSTANTS_H
#define ZXC_CONSTANTS_H

/* library this build is drop-in compatible with: reference v0.13.3, wire format v8 */
#define ZXC_VERSION_MAJOR 0
#define ZXC_VERSION_MINOR 13
#define ZXC_VERSION_PATCH 3
#define ZXC_LIB_VERSION_STR "0.13.3"

#define ZXC_BLOCK_SIZE_MIN_LOG2 12            /* 4 KiB  (include/zxc_constants.h:56) */
#define ZXC_BLOCK_SIZE_MAX_LOG2 21            /* 2 MiB */
#define ZXC_BLOCK_SIZE_DEFAULT (512 * 1024)
#define ZXC_BLOCK_SIZE_MIN (1U << ZXC_BLOCK_SIZE_MIN_LOG2)
#define ZXC_BLOCK_SIZE_MAX (1U << ZXC_BLOCK_SIZE_MAX_LOG2)
#define ZXC_DICT_SIZE_MAX ((1U << 16) - 1U)
#define ZXC_HUF_TABLE_SIZE 128
#define ZXC_MAX_THREADS 512
#define ZXC_FILE_HEADER_SIZE 16
#define ZXC_FILE_FOOTER_SIZE 12

typedef enum {
    ZXC_LEVEL_FASTEST = 1,
    ZXC_LEVEL_FAST = 2,
    ZXC_LEVEL_DEFAULT = 3,
    ZXC_LEVEL_BALANCED = 4,
    ZXC_LEVEL_COMPACT = 5,
    ZXC_LEVEL_DENSITY = 6,
    ZXC_LEVEL_ULTRA = 7
} zxc_compression_level_t;

#endif
