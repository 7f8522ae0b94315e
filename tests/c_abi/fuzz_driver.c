/* fuzz_driver.c — TEST INFRASTRUCTURE: a deterministic driver for the reference's own libFuzzer harnesses
 * (/root/reference/tests/fuzz_{roundtrip,decompress,seekable,pstream,dict}.c, compiled in place with their asserts on, see the
 * Makefile) linked against libzxc_mi355x.so (or the mock-device build of its host sources). There is no libFuzzer here, so the inputs
 * are made up: random bytes, skewed bytes, repetitive text, and — for the harnesses that parse archives — valid archives of such data
 * with a few bytes stomped. A harness reports a problem by assert() -> abort(). Usage: fuzz_<name> [iterations] [seed]. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/zxc.h"

int LLVMFuzzerTestOneInput(const uint8_t* data, size_t size);

static uint64_t g_s;
static uint64_t rnd(void) { /* splitmix64 */
    uint64_t z = (g_s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static size_t gen(uint8_t* b, size_t cap) {
    static const size_t sizes[] = {0, 1, 2, 3, 4, 17, 255, 4095, 4096, 4097, 20000, 65536, 70000, 200000, 600000};
    size_t n = sizes[rnd() % (sizeof sizes / sizeof sizes[0])];
    if (rnd() % 3 == 0) n = (size_t)(rnd() % 100000);
    if (n > cap) n = cap;
    switch (rnd() % 4) {
        case 0: for (size_t i = 0; i < n; i++) b[i] = (uint8_t)rnd(); break;
        case 1: for (size_t i = 0; i < n; i++) b[i] = (uint8_t)(rnd() & 7); break;
        case 2: { /* words from a small vocabulary */
            static const char* w[] = {"alpha ", "beta ", "gamma,", "delta\n", "{\"id\": ", "\"status\": \"active\"", "0123456789", "    "};
            size_t i = 0;
            while (i < n) { const char* s = w[rnd() % 8]; for (; *s && i < n; s++) b[i++] = (uint8_t)*s; }
            break;
        }
        default: { /* runs and a little noise */
            size_t i = 0;
            while (i < n) { const uint8_t v = (uint8_t)rnd(); size_t k = 1 + rnd() % 300; for (; k && i < n; k--) b[i++] = v; if (i < n && rnd() % 2) b[i++] = (uint8_t)rnd(); }
        }
    }
    return n;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    g_s = argc > 2 ? strtoull(argv[2], NULL, 0) : 1;
    const size_t cap = 700000;
    uint8_t* in = (uint8_t*)malloc(cap + 64);
    uint8_t* arc = (uint8_t*)malloc((size_t)zxc_compress_bound(cap) + 64);
    if (!in || !arc) return 2;
    int archives = 0;
    for (int it = 0; it < iters; it++) {
        size_t n = gen(in, cap);
        if (it % 2 == 1 && n > 0) { /* a valid archive of it, some bytes stomped: food for the parsers */
            zxc_compress_opts_t o;
            memset(&o, 0, sizeof o);
            o.level = 1 + (int)(rnd() % 5);
            o.block_size = (size_t)4096 << (rnd() % 6);
            o.checksum_enabled = (int)(rnd() & 1);
            o.seekable = (int)(rnd() & 1);
            const int64_t c = zxc_compress(in, n, arc, (size_t)zxc_compress_bound(n), &o);
            if (c > 0) {
                const int stomps = (int)(rnd() % 4);
                for (int k = 0; k < stomps; k++) arc[rnd() % (uint64_t)c] ^= (uint8_t)(1u << (rnd() % 8));
                size_t len = (size_t)c;
                if (rnd() % 8 == 0) len = (size_t)(rnd() % (uint64_t)c);
                LLVMFuzzerTestOneInput(arc, len);
                archives++;
                continue;
            }
        }
        LLVMFuzzerTestOneInput(in, n);
    }
    printf("FUZZ OK %d inputs (%d mutated archives)\n", iters, archives);
    free(in);
    free(arc);
    return 0;
}
