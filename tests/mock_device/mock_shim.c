/* mock_shim.c — TEST INFRASTRUCTURE: a "device" made of host memory, so that the HOST side of libzxc_mi355x.so (zxc_host.c and what
 * it includes: the piece pipeline, the FILE* and push stream callers, error precedence, framing) can run in `-m "not gpu"` tests on
 * a box without a GPU. It implements the entry points zxc_host.c expects from zxc_hip_shim.hip: memory = malloc, copies = memcpy,
 * streams / events = nothing (everything is synchronous), and the two kernels' contracts — "decode this table of independent
 * blocks", "encode these blocks into slots" — by calling the UNMODIFIED reference's Block API (oracle/_ref/libzxc_ref.so, found
 * with dlopen so that its zxc_* names do not collide with the product's). Never shipped, never loaded by the product: the
 * product library links zxc_hip_shim.hip and fails with ZXC_ERROR_GPU_UNAVAILABLE without a HIP device. Built by ./Makefile into
 * _bin/libzxc_mockdev.so. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/zxc.h"

typedef void* (*create_cctx_fn)(const zxc_compress_opts_t*);
typedef void* (*create_dctx_fn)(void);
typedef void (*free_ctx_fn)(void*);
typedef int64_t (*compress_block_fn)(void*, const void*, size_t, void*, size_t, const zxc_compress_opts_t*);
typedef int64_t (*decompress_block_fn)(void*, const void*, size_t, void*, size_t, const zxc_decompress_opts_t*);
static struct {
    void* h;
    create_cctx_fn create_cctx;
    create_dctx_fn create_dctx;
    free_ctx_fn free_cctx, free_dctx;
    compress_block_fn compress_block;
    decompress_block_fn decompress_block;
} R;

static void ref_load_once(void);
static pthread_once_t g_ref_once = PTHREAD_ONCE_INIT;
static int ref_load(void) {
    pthread_once(&g_ref_once, ref_load_once);
    return R.h != NULL;
}
static void ref_load_once(void) {
    const char* p = getenv("ZXC_MOCK_REF_SO");
    Dl_info info;
    char path[4096];
    if (!p && dladdr((void*)ref_load_once, &info) && info.dli_fname) { /* <repo>/tests/mock_device/_bin/x.so -> <repo>/oracle/_ref/libzxc_ref.so */
        snprintf(path, sizeof path, "%s", info.dli_fname);
        char* s = strrchr(path, '/');
        if (s) { *s = 0; snprintf(s, sizeof path - (size_t)(s - path), "/../../../oracle/_ref/libzxc_ref.so"); p = path; }
    }
    void* h = p ? dlopen(p, RTLD_NOW | RTLD_LOCAL) : NULL;
    if (!h) return;
    R.create_cctx = (create_cctx_fn)dlsym(h, "zxc_create_cctx");
    R.create_dctx = (create_dctx_fn)dlsym(h, "zxc_create_dctx");
    R.free_cctx = (free_ctx_fn)dlsym(h, "zxc_free_cctx");
    R.free_dctx = (free_ctx_fn)dlsym(h, "zxc_free_dctx");
    R.compress_block = (compress_block_fn)dlsym(h, "zxc_compress_block");
    R.decompress_block = (decompress_block_fn)dlsym(h, "zxc_decompress_block");
    if (!R.create_cctx || !R.create_dctx || !R.compress_block || !R.decompress_block) return;
    R.h = h;
}

#define EXPORT __attribute__((visibility("default")))
static __thread int g_dev = 0;
static int mock_devices(void) { /* ZXC_MOCK_DEVICES=0: behave like a box without a device */
    const char* e = getenv("ZXC_MOCK_DEVICES");
    return e ? atoi(e) : 1;
}

EXPORT int zxc_mi355x_device_count(void) {
    const int n = mock_devices();
    return n > 0 && ref_load() ? n : 0;
}
EXPORT int zxc_mi355x_set_device(int d) { if (d < 0 || d >= mock_devices()) return ZXC_ERROR_GPU_UNAVAILABLE; g_dev = d; return ZXC_OK; }
EXPORT int zxc_mi355x_get_device(void) { return g_dev; }
int zxc_hip_current_device(void) { return g_dev; }
EXPORT void* zxc_mi355x_malloc(size_t n) { return malloc(n ? n : 16); }
EXPORT void zxc_mi355x_free(void* p) { free(p); }
EXPORT int zxc_mi355x_memcpy_h2d(void* d, const void* s, size_t n) { if (n) memcpy(d, s, n); return ZXC_OK; }
EXPORT int zxc_mi355x_memcpy_d2h(void* d, const void* s, size_t n) { if (n) memcpy(d, s, n); return ZXC_OK; }
EXPORT int zxc_mi355x_synchronize(void* stream) { (void)stream; return ZXC_OK; }
EXPORT void zxc_mi355x_release_cached(void) {}
int zxc_hip_stream_create(void** s) { *s = malloc(1); return *s ? ZXC_OK : ZXC_ERROR_MEMORY; }
void zxc_hip_stream_destroy(void* s) { free(s); }
int zxc_hip_memcpy_h2d_async(void* d, const void* s, size_t n, void* st) { (void)st; if (n) memcpy(d, s, n); return ZXC_OK; }
int zxc_hip_memcpy_d2h_async(void* d, const void* s, size_t n, void* st) { (void)st; if (n) memcpy(d, s, n); return ZXC_OK; }
int zxc_hip_event_create(void** e) { *e = malloc(1); return *e ? ZXC_OK : ZXC_ERROR_MEMORY; }
void zxc_hip_event_destroy(void* e) { free(e); }
int zxc_hip_event_record(void* e, void* s) { (void)e; (void)s; return ZXC_OK; }
int zxc_hip_event_synchronize(void* e) { (void)e; return ZXC_OK; }
void* zxc_hip_host_alloc(size_t n) { return malloc(n ? n : 16); }
void zxc_hip_host_free(void* p) { free(p); }

/* the decode kernel's contract (include/zxc_mi355x.h:62-72): status = decoded size or a negative zxc_error_t; the first out_len
 * bytes are kept at out_off */
int zxc_hip_decode_blocks(const void* d_comp, const zxc_dev_job_t* jobs, uint32_t n, void* d_out, int32_t* st, uint32_t block_size,
                          int verify, const void* d_dict, uint32_t dict_size, const void* d_dict_huf, uint32_t cap_override, void* stream) {
    (void)stream;
    if (!ref_load()) return ZXC_ERROR_GPU_UNAVAILABLE;
    const size_t cap = cap_override ? cap_override : (size_t)block_size + 2112u;
    uint8_t* tmp = (uint8_t*)malloc(cap + 64);
    void* dctx = R.create_dctx();
    if (!tmp || !dctx) { free(tmp); return ZXC_ERROR_MEMORY; }
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t* src = (const uint8_t*)d_comp + jobs[i].comp_off;
        size_t sz = jobs[i].comp_size;
        if (sz >= 8) { /* the block itself: header + payload (+ trailer when it is to be verified) */
            const uint64_t phys = 8ull + ((uint32_t)src[3] | ((uint32_t)src[4] << 8) | ((uint32_t)src[5] << 16) | ((uint32_t)src[6] << 24)) + (verify ? 4u : 0u);
            if (phys < sz) sz = (size_t)phys;
        }
        zxc_decompress_opts_t o;
        memset(&o, 0, sizeof o);
        o.checksum_enabled = verify;
        if (dict_size) { o.dict = d_dict; o.dict_size = dict_size; o.dict_huf = d_dict_huf; }
        const int64_t r = R.decompress_block(dctx, src, sz, tmp, cap, &o);
        st[i] = (int32_t)r;
        if (r > 0) memcpy((uint8_t*)d_out + jobs[i].out_off, tmp, (size_t)r < jobs[i].out_len ? (size_t)r : jobs[i].out_len);
    }
    if (R.free_dctx) R.free_dctx(dctx);
    free(tmp);
    return ZXC_OK;
}
EXPORT int zxc_mi355x_decode_blocks_device(const void* d_comp, const zxc_dev_job_t* jobs, uint32_t n, void* d_out, int32_t* st,
                                           uint32_t block_size, int verify, void* stream) {
    return zxc_hip_decode_blocks(d_comp, jobs, n, d_out, st, block_size, verify, NULL, 0, NULL, 0, stream);
}

/* the encode kernel's contract (include/zxc_mi355x.h:83-94) */
EXPORT uint32_t zxc_mi355x_encode_slot_stride(uint32_t block_size) { return 2u * block_size + 512u; }
EXPORT int zxc_mi355x_encode_blocks_device(const void* d_src, uint64_t src_size, uint32_t block_size, int level, int with_checksum,
                                           void* d_slots, uint32_t* d_sizes, void* stream) {
    (void)stream;
    if (!ref_load()) return ZXC_ERROR_GPU_UNAVAILABLE;
    zxc_compress_opts_t o;
    memset(&o, 0, sizeof o);
    o.level = level;
    o.block_size = block_size;
    o.checksum_enabled = with_checksum;
    void* cctx = R.create_cctx(&o);
    if (!cctx) return ZXC_ERROR_MEMORY;
    const uint32_t stride = zxc_mi355x_encode_slot_stride(block_size);
    const uint64_t nb = (src_size + block_size - 1) / block_size;
    int rc = ZXC_OK;
    for (uint64_t i = 0; i < nb && rc == ZXC_OK; i++) {
        const uint64_t o0 = i * block_size, len = src_size - o0 < block_size ? src_size - o0 : block_size;
        const int64_t r = R.compress_block(cctx, (const uint8_t*)d_src + o0, (size_t)len, (uint8_t*)d_slots + i * stride, stride, &o);
        if (r < 0) rc = (int)r;
        else d_sizes[i] = (uint32_t)r;
    }
    if (R.free_cctx) R.free_cctx(cctx);
    return rc;
}
EXPORT uint64_t zxc_mi355x_encode_dict_work_size(uint64_t src_size, uint32_t block_size, uint32_t dict_size) {
    (void)src_size; (void)block_size; (void)dict_size;
    return 16;
}
/* with a dictionary (include/zxc_mi355x.h:95-102): the reference's Block API takes the dictionary in its options */
EXPORT int zxc_mi355x_encode_blocks_dict_device(const void* d_src, uint64_t src_size, uint32_t block_size, int level, int ck, const void* d_dict,
                                                uint32_t dict_size, void* d_work, void* d_slots, uint32_t* d_sizes, void* stream) {
    (void)d_work; (void)stream;
    if (!ref_load()) return ZXC_ERROR_GPU_UNAVAILABLE;
    zxc_compress_opts_t o;
    memset(&o, 0, sizeof o);
    o.level = level;
    o.block_size = block_size;
    o.checksum_enabled = ck;
    o.dict = d_dict;
    o.dict_size = dict_size;
    void* cctx = R.create_cctx(&o);
    if (!cctx) return ZXC_ERROR_MEMORY;
    const uint32_t stride = zxc_mi355x_encode_slot_stride(block_size);
    const uint64_t nb = (src_size + block_size - 1) / block_size;
    int rc = ZXC_OK;
    for (uint64_t i = 0; i < nb && rc == ZXC_OK; i++) {
        const uint64_t o0 = i * block_size, len = src_size - o0 < block_size ? src_size - o0 : block_size;
        const int64_t r = R.compress_block(cctx, (const uint8_t*)d_src + o0, (size_t)len, (uint8_t*)d_slots + i * stride, stride, &o);
        if (r < 0) rc = (int)r;
        else d_sizes[i] = (uint32_t)r;
    }
    if (R.free_cctx) R.free_cctx(cctx);
    return rc;
}
int zxc_hip_block_offsets(uint32_t* sizes, uint64_t* offs, uint32_t n, uint32_t max_size, void* stream) {
    (void)stream;
    uint64_t t = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (sizes[i] > max_size) sizes[i] = max_size;
        offs[i] = t;
        t += sizes[i];
    }
    return ZXC_OK;
}
EXPORT int zxc_mi355x_gather_blocks_device(const void* d_slots, uint32_t block_size, const uint32_t* sizes, const uint64_t* offs, void* d_out,
                                           uint32_t n, void* stream) {
    (void)stream;
    const uint32_t stride = zxc_mi355x_encode_slot_stride(block_size);
    for (uint32_t i = 0; i < n; i++) memcpy((uint8_t*)d_out + offs[i], (const uint8_t*)d_slots + (size_t)i * stride, sizes[i]);
    return ZXC_OK;
}
