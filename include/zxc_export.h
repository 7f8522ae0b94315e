/* zxc_export.h — symbol visibility for libzxc_mi355x.so.
 * Replaces reference include/zxc_export.h:37-91 (ZXC_EXPORT on every public symbol). */
#ifndef ZXC_EXPORT_H
#define ZXC_EXPORT_H
#if defined(__GNUC__) || defined(__clang__)
#define ZXC_EXPORT __attribute__((visibility("default")))
#else
#define ZXC_EXPORT
#endif
#endif
