// wave_emu.cpp — fiber scheduler of the CPU wave emulator (see wave_emu.h). Test infrastructure.
#include "wave_emu.h"

#include <stdio.h>
#include <stdlib.h>
#include <ucontext.h>

#include <functional>
#include <vector>

namespace emu {
namespace {
constexpr size_t STACK = 1u << 20;
constexpr int MAXW = 8;  // wavefronts per workgroup (512 threads)
struct WLog { uintptr_t addr; uint8_t size; bool is_or; uint8_t data[16]; };
struct WaveSt {
    ucontext_t ctx[W];
    char* stack[W] = {};
    bool done[W];
    int live = 0, arrived = 0;
    unsigned gen = 0;
    uint64_t in[W], snap[2][W], act[2], arrived_mask = 0;
    int tag[W];
    // buffered LDS writes (emu_lds_hooks.h): per lane, applied at the wave's next rendezvous
    std::vector<WLog> wlog[W];
};
struct State {
    ucontext_t sched;
    WaveSt wv[MAXW];
    int n_waves = 1, cw = 0, cur = 0;
    int wg_live = 0, wg_arrived = 0;  // workgroup barrier (s_barrier): lanes of all waves
    unsigned wg_gen = 0;
    Idx bid{0, 0, 0}, gdim{1, 1, 1};
    std::function<void()> body;
    uint64_t clk = 0;
    std::vector<uint8_t> shadow;  // per LDS byte of this interval: 0 = untouched, 1..64 = plain store by lane-1, 65 = or
} S;
extern "C" char __start_emu_lds[], __stop_emu_lds[];

void commit_lds(WaveSt& V) {
    const uintptr_t lo = (uintptr_t)__start_emu_lds, hi = (uintptr_t)__stop_emu_lds;
    if (S.shadow.size() != hi - lo) S.shadow.assign(hi - lo, 0);
    std::vector<uintptr_t> touched;
    for (int l = 0; l < W; l++) {
        for (const auto& w : V.wlog[l]) {
            if (w.addr < lo || w.addr + w.size > hi) { fprintf(stderr, "wave_emu: LDS hook on a non-LDS address\n"); abort(); }
            for (unsigned k = 0; k < w.size; k++) {
                uint8_t& sh = S.shadow[w.addr + k - lo];
                uint8_t* m = (uint8_t*)(w.addr + k);
                if (w.is_or) {
                    if (sh >= 1 && sh <= 64 && sh != l + 1) {
                        fprintf(stderr, "wave_emu: ds_or of lane %d meets a plain store of lane %d on one LDS byte without a fence\n", l, sh - 1);
                        abort();
                    }
                    *m |= w.data[k];
                    if (sh == 0) { sh = 65; touched.push_back(w.addr + k - lo); }
                } else {
                    if (sh != 0 && sh != l + 1 && !(sh <= 64 && *m == w.data[k])) {
                        fprintf(stderr, "wave_emu: lanes %d and %d write one LDS byte (offset %zu) in one interval without a fence\n",
                                sh == 65 ? -1 : sh - 1, l, (size_t)(w.addr + k - lo));
                        abort();
                    }
                    *m = w.data[k];
                    if (sh == 0) touched.push_back(w.addr + k - lo);
                    sh = (uint8_t)(l + 1);
                }
            }
        }
        V.wlog[l].clear();
    }
    for (uintptr_t t : touched) S.shadow[t] = 0;
}

void release(WaveSt& V) {
    commit_lds(V);
    const unsigned g = V.gen + 1u;
    int t = -1;
    for (int i = 0; i < W; i++) {
        if ((V.arrived_mask >> i) & 1) {
            if (t < 0) t = V.tag[i];
            else if (t != V.tag[i]) {
                fprintf(stderr, "wave_emu: lanes meet at different cross-lane operations (source lines %d and %d): "
                                "a cross-lane operation sits inside divergent control flow\n", t, V.tag[i]);
                abort();
            }
            V.snap[g & 1][i] = V.in[i];
        } else {
            V.snap[g & 1][i] = 0;
        }
    }
    V.act[g & 1] = V.arrived_mask;
    V.arrived = 0;
    V.arrived_mask = 0;
    V.gen = g;
}

void trampoline() {
    S.body();
    WaveSt& V = S.wv[S.cw];
    V.done[S.cur] = true;
    V.live--;
    S.wg_live--;
    if (V.live == 0) commit_lds(V);
    if (V.arrived > 0 && V.arrived == V.live) release(V);
    if (S.wg_arrived > 0 && S.wg_arrived == S.wg_live) { S.wg_arrived = 0; S.wg_gen++; }  // (s_barrier does not wait for ended waves)
    swapcontext(&V.ctx[S.cur], &S.sched);
}
}  // namespace

void lds_write(void* p, const void* data, unsigned size, bool is_or) {
    WLog w;
    w.addr = (uintptr_t)p;
    w.size = (uint8_t)size;
    w.is_or = is_or;
    memcpy(w.data, data, size);
    if ((w.addr & (size - 1u)) != 0) { fprintf(stderr, "wave_emu: misaligned %u-byte LDS store\n", size); abort(); }
    S.wv[S.cw].wlog[S.cur].push_back(w);
}
void lds_read(const void* p, void* out, unsigned size) {
    if (((uintptr_t)p & (size - 1u)) != 0) { fprintf(stderr, "wave_emu: misaligned %u-byte LDS load\n", size); abort(); }
    lds_read_unaligned(p, out, size);
}
void lds_read_unaligned(const void* p, void* out, unsigned size) {  // (gfx950 runs the LDS in unaligned access mode: any byte address)
    const uintptr_t a = (uintptr_t)p;
    if (a < (uintptr_t)__start_emu_lds || a + size > (uintptr_t)__stop_emu_lds) { fprintf(stderr, "wave_emu: LDS load outside the LDS\n"); abort(); }
    memcpy(out, p, size);
    uint8_t* o = (uint8_t*)out;
    for (const auto& w : S.wv[S.cw].wlog[S.cur]) {  // the lane's own earlier writes of this interval, in order
        if (w.addr + w.size <= a || a + size <= w.addr) continue;
        for (unsigned k = 0; k < w.size; k++) {
            const uintptr_t b = w.addr + k;
            if (b < a || b >= a + size) continue;
            if (w.is_or) o[b - a] |= w.data[k];
            else o[b - a] = w.data[k];
        }
    }
}
int lane() { return S.cur; }
Idx tid() { return Idx{(unsigned)(S.cw * W + S.cur), 0, 0}; }
Idx bid() { return S.bid; }
Idx gdim() { return S.gdim; }
uint64_t clock64() { return S.clk += 7; }
void fail(const char* what, int tag) {
    fprintf(stderr, "wave_emu: %s (%d)\n", what, tag);
    abort();
}

const uint64_t* sync(uint64_t v, int tag, uint64_t* active) {
    const int me = S.cur, w = S.cw;
    WaveSt& V = S.wv[w];
    V.in[me] = v;
    V.tag[me] = tag;
    V.arrived_mask |= 1ull << me;
    V.arrived++;
    const unsigned g = V.gen;
    if (V.arrived == V.live) release(V);
    while (V.gen == g) swapcontext(&V.ctx[me], &S.sched);
    // a lane can be at most one rendezvous ahead of the slowest one of its wave, so generation g+1 is still intact
    *active = V.act[(g + 1u) & 1];
    return V.snap[(g + 1u) & 1];
}

// s_barrier / __syncthreads: a rendezvous of the lane's own wave first (its LDS writes are applied, divergence inside the wave
// is caught), then of every live lane of the workgroup. Waves may arrive from different call sites, as on the hardware.
void wg_barrier(int tag) {
    barrier(tag);
    if (S.n_waves == 1) return;
    const int me = S.cur, w = S.cw;
    const unsigned g = S.wg_gen;
    S.wg_arrived++;
    if (S.wg_arrived == S.wg_live) { S.wg_arrived = 0; S.wg_gen++; }
    while (S.wg_gen == g) swapcontext(&S.wv[w].ctx[me], &S.sched);
}

// Runs `body` once per thread of one workgroup of n_lanes <= 512 threads (wavefronts of 64) as block `block` of `grid`.
void run_wave(const std::function<void()>& body, unsigned block, unsigned grid, int n_lanes) {
    const int nw = (n_lanes + W - 1) / W;
    if (nw < 1 || nw > MAXW) fail("workgroup size", n_lanes);
    S.body = body;
    S.bid = Idx{block, 0, 0};
    S.gdim = Idx{grid, 1, 1};
    S.n_waves = nw;
    S.wg_live = n_lanes;
    S.wg_arrived = 0;
    for (int w = 0; w < nw; w++) {
        WaveSt& V = S.wv[w];
        const int nl = w + 1 < nw ? W : n_lanes - W * (nw - 1);
        V.live = nl;
        V.arrived = 0;
        V.arrived_mask = 0;
        for (int i = 0; i < W; i++) V.done[i] = i >= nl;
        for (int i = 0; i < nl; i++) {
            if (!V.stack[i]) V.stack[i] = (char*)malloc(STACK);
            getcontext(&V.ctx[i]);
            V.ctx[i].uc_stack.ss_sp = V.stack[i];
            V.ctx[i].uc_stack.ss_size = STACK;
            V.ctx[i].uc_link = &S.sched;
            makecontext(&V.ctx[i], (void (*)())trampoline, 0);
        }
    }
    for (;;) {
        bool any = false;
        for (int w = 0; w < nw; w++) {
            WaveSt& V = S.wv[w];
            for (int i = 0; i < W; i++) {
                if (V.done[i]) continue;
                any = true;
                S.cw = w;
                S.cur = i;
                swapcontext(&S.sched, &V.ctx[i]);
            }
        }
        if (!any) break;
    }
}
}  // namespace emu
