/* huf_helpers.c — TEST INFRASTRUCTURE for tests/c_abi: the reference's tests/test_dict.c builds a test table for its .zxd cases
 * with two INTERNAL helpers of the reference library (zxc_huf_build_code_lengths, zxc_huf_pack_lengths: src/lib/zxc_huffman.c:172,
 * :952 — not part of the public API, not exported by either library). The harness provides them: length-limited Huffman lengths by
 * package-merge (lists of (weight, symbol multiset) per round) and the 256 x 4-bit packing (low nibble first). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t w; uint8_t cnt[256]; } item_t;
static int by_weight(const void* a, const void* b) {
    const uint64_t* x = (const uint64_t*)a;
    const uint64_t* y = (const uint64_t*)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    return x[1] < y[1] ? -1 : 1;
}
int zxc_huf_build_code_lengths(const uint32_t* freq, uint8_t* code_len, void* scratch, const int max_code_len) {
    (void)scratch;
    memset(code_len, 0, 256);
    uint64_t leaf[256][2];
    int n = 0;
    for (int s = 0; s < 256; s++)
        if (freq[s]) { leaf[n][0] = freq[s]; leaf[n][1] = (uint64_t)s; n++; }
    if (n == 0) return -8;
    if (n == 1) { code_len[leaf[0][1]] = 1; return 0; }
    qsort(leaf, (size_t)n, sizeof leaf[0], by_weight);
    item_t* prev = (item_t*)calloc((size_t)2 * n, sizeof(item_t));
    item_t* cur = (item_t*)calloc((size_t)2 * n, sizeof(item_t));
    if (!prev || !cur) { free(prev); free(cur); return -1; }
    int n_prev = 0;
    for (int round = 0; round < max_code_len; round++) {
        const int packs = n_prev / 2;
        int li = 0, pi = 0, k = 0;
        while (li < n || pi < packs) {
            const uint64_t pw = pi < packs ? prev[2 * pi].w + prev[2 * pi + 1].w : 0;
            if (pi >= packs || (li < n && leaf[li][0] <= pw)) {
                memset(cur[k].cnt, 0, 256);
                cur[k].w = leaf[li][0];
                cur[k].cnt[leaf[li][1]] = 1;
                li++;
            } else {
                cur[k].w = pw;
                for (int s = 0; s < 256; s++) cur[k].cnt[s] = (uint8_t)(prev[2 * pi].cnt[s] + prev[2 * pi + 1].cnt[s]);
                pi++;
            }
            k++;
        }
        item_t* t = prev; prev = cur; cur = t;
        n_prev = k;
    }
    for (int i = 0; i < 2 * n - 2 && i < n_prev; i++)
        for (int s = 0; s < 256; s++) code_len[s] = (uint8_t)(code_len[s] + prev[i].cnt[s]);
    free(prev);
    free(cur);
    return 0;
}
void zxc_huf_pack_lengths(const uint8_t* code_len, uint8_t* out) {
    for (int i = 0; i < 256; i += 2) out[i >> 1] = (uint8_t)((code_len[i] & 0x0F) | ((code_len[i + 1] & 0x0F) << 4));
}
