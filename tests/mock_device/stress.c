/* stress.c — TEST INFRASTRUCTURE: four threads drive the host API over the mock device at once (frames, push streams in odd
 * chunkings, seekable ranges incl. _mt), with 1 MiB pieces so that every call runs the piece pipeline with its producer threads.
 * Built with -fsanitize=thread or =address,undefined by tests/test_sanitizers.py (the product's host sources are compiled into the
 * same instrumented library): the races and memory errors of the HOST code show up here, on a box without a GPU. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/zxc.h"

static uint64_t mix(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void fill(uint8_t* b, size_t n, uint64_t* s) {
    static const char* w[] = {"alpha ", "beta ", "gamma,", "delta\n", "{\"id\": ", "\"status\": \"active\"", "0123456789", "    "};
    size_t i = 0;
    while (i < n) {
        if (mix(s) % 7 == 0) { size_t k = 1 + mix(s) % 40; for (; k && i < n; k--) b[i++] = (uint8_t)mix(s); continue; }
        const char* t = w[mix(s) % 8];
        for (; *t && i < n; t++) b[i++] = (uint8_t)*t;
    }
}
static int g_iters = 6;
static int g_failed = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "stress: %s failed (thread %d, line %d)\n", #c, id, __LINE__); __atomic_store_n(&g_failed, 1, __ATOMIC_SEQ_CST); goto out; } } while (0)

static void* worker(void* arg) {
    const int id = (int)(intptr_t)arg;
    uint64_t s = 1000 + (uint64_t)id;
    const size_t cap = 6u << 20;
    uint8_t* data = (uint8_t*)malloc(cap);
    uint8_t* back = (uint8_t*)malloc(cap + 4096);
    const size_t acap = (size_t)zxc_compress_bound(cap);
    uint8_t* arc = (uint8_t*)malloc(acap);
    uint8_t* arc2 = (uint8_t*)malloc(acap);
    if (!data || !back || !arc || !arc2) return NULL;
    for (int it = 0; it < g_iters; it++) {
        const size_t n = 1 + (size_t)(mix(&s) % cap);
        fill(data, n, &s);
        zxc_compress_opts_t co;
        memset(&co, 0, sizeof co);
        co.level = 1 + (int)(mix(&s) % 5);
        co.block_size = (size_t)4096 << (mix(&s) % 3);
        co.checksum_enabled = (int)(mix(&s) & 1);
        co.seekable = 1;
        zxc_decompress_opts_t dopt;
        memset(&dopt, 0, sizeof dopt);
        dopt.checksum_enabled = co.checksum_enabled;
        const int64_t c = zxc_compress(data, n, arc, acap, &co);
        CHECK(c > 0);
        CHECK(zxc_decompress(arc, (size_t)c, back, n, &dopt) == (int64_t)n && memcmp(back, data, n) == 0);
        /* seekable ranges */
        zxc_seekable* sk = zxc_seekable_open(arc, (size_t)c);
        CHECK(sk != NULL);
        for (int r = 0; r < 3; r++) {
            const size_t off = (size_t)(mix(&s) % n), len = 1 + (size_t)(mix(&s) % (n - off));
            const int64_t g = r & 1 ? zxc_seekable_decompress_range_mt(sk, back, len, off, len, 3) : zxc_seekable_decompress_range(sk, back, len, off, len);
            if (g != (int64_t)len || memcmp(back, data + off, len) != 0) { zxc_seekable_free(sk); CHECK(!"seekable range"); }
        }
        zxc_seekable_free(sk);
        /* push streams: the archive must be zxc_compress's non-seekable one; then read it back */
        co.seekable = 0;
        const int64_t c0 = zxc_compress(data, n, arc, acap, &co);
        CHECK(c0 > 0);
        zxc_cstream* cs = zxc_cstream_create(&co);
        CHECK(cs != NULL);
        const size_t in_chunk = 1 + (size_t)(mix(&s) % (2u << 20)), out_chunk = 1 + (size_t)(mix(&s) % (3u << 20));
        size_t got = 0, fed = 0;
        int bad = 0;
        while (fed < n && !bad) {
            zxc_inbuf_t in = {data + fed, n - fed < in_chunk ? n - fed : in_chunk, 0};
            while (in.pos < in.size && !bad) {
                zxc_outbuf_t out = {arc2 + got, acap - got < out_chunk ? acap - got : out_chunk, 0};
                if (zxc_cstream_compress(cs, &out, &in) < 0) bad = 1;
                got += out.pos;
            }
            fed += in.size;
        }
        for (int64_t p = 1; p > 0 && !bad;) {
            zxc_outbuf_t out = {arc2 + got, acap - got < out_chunk ? acap - got : out_chunk, 0};
            p = zxc_cstream_end(cs, &out);
            if (p < 0) bad = 1;
            got += out.pos;
        }
        zxc_cstream_free(cs);
        CHECK(!bad && got == (size_t)c0 && memcmp(arc, arc2, got) == 0);
        zxc_dstream* ds = zxc_dstream_create(&dopt);
        CHECK(ds != NULL);
        size_t dgot = 0, dfed = 0;
        while (!zxc_dstream_finished(ds) && !bad) {
            zxc_inbuf_t in = {arc2 + dfed, got - dfed < in_chunk ? got - dfed : in_chunk, 0};
            zxc_outbuf_t out = {back + dgot, n - dgot < out_chunk ? n - dgot : out_chunk, 0};
            const int64_t r = zxc_dstream_decompress(ds, &out, &in);
            if (r < 0 || (r == 0 && in.pos == 0 && out.pos == 0 && !zxc_dstream_finished(ds) && dfed >= got)) bad = 1;
            dgot += out.pos;
            dfed += in.pos;
        }
        zxc_dstream_free(ds);
        CHECK(!bad && dgot == n && memcmp(back, data, n) == 0);
    }
out:
    free(data); free(back); free(arc); free(arc2);
    return NULL;
}

int main(int argc, char** argv) {
    if (argc > 1) g_iters = atoi(argv[1]);
    pthread_t th[4];
    for (int i = 0; i < 4; i++) pthread_create(&th[i], NULL, worker, (void*)(intptr_t)i);
    for (int i = 0; i < 4; i++) pthread_join(th[i], NULL);
    zxc_mi355x_release_cached();
    printf(g_failed ? "STRESS FAILED\n" : "STRESS OK\n");
    return g_failed;
}
