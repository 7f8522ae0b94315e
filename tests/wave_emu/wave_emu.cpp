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
struct State {
    ucontext_t sched;
    ucontext_t ctx[W];
    char* stack[W];
    bool done[W];
    int cur = 0, live = 0, arrived = 0;
    unsigned gen = 0;
    uint64_t in[W], snap[2][W], act[2], arrived_mask = 0;
    int tag[W];
    Idx bid{0, 0, 0}, gdim{1, 1, 1};
    int n_lanes = W;
    std::function<void()> body;
    uint64_t clk = 0;
    bool stacks_ready = false;
    // buffered LDS writes (emu_lds_hooks.h): per lane, applied at the next rendezvous
    struct WLog { uintptr_t addr; uint8_t size; bool is_or; uint8_t data[16]; };
    std::vector<WLog> wlog[W];
    std::vector<uint8_t> shadow;  // per LDS byte of this interval: 0 = untouched, 1..64 = plain store by lane-1, 65 = or
} S;
extern "C" char __start_emu_lds[], __stop_emu_lds[];

void commit_lds() {
    const uintptr_t lo = (uintptr_t)__start_emu_lds, hi = (uintptr_t)__stop_emu_lds;
    if (S.shadow.size() != hi - lo) S.shadow.assign(hi - lo, 0);
    std::vector<uintptr_t> touched;
    for (int l = 0; l < W; l++) {
        for (const auto& w : S.wlog[l]) {
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
        S.wlog[l].clear();
    }
    for (uintptr_t t : touched) S.shadow[t] = 0;
}

void release() {
    commit_lds();
    const unsigned g = S.gen + 1u;
    int t = -1;
    for (int i = 0; i < W; i++) {
        if ((S.arrived_mask >> i) & 1) {
            if (t < 0) t = S.tag[i];
            else if (t != S.tag[i]) {
                fprintf(stderr, "wave_emu: lanes meet at different cross-lane operations (source lines %d and %d): "
                                "a cross-lane operation sits inside divergent control flow\n", t, S.tag[i]);
                abort();
            }
            S.snap[g & 1][i] = S.in[i];
        } else {
            S.snap[g & 1][i] = 0;
        }
    }
    S.act[g & 1] = S.arrived_mask;
    S.arrived = 0;
    S.arrived_mask = 0;
    S.gen = g;
}

void trampoline() {
    S.body();
    S.done[S.cur] = true;
    S.live--;
    if (S.live == 0) commit_lds();
    if (S.arrived > 0 && S.arrived == S.live) release();
    swapcontext(&S.ctx[S.cur], &S.sched);
}
}  // namespace

void lds_write(void* p, const void* data, unsigned size, bool is_or) {
    State::WLog w;
    w.addr = (uintptr_t)p;
    w.size = (uint8_t)size;
    w.is_or = is_or;
    memcpy(w.data, data, size);
    if ((w.addr & (size - 1u)) != 0) { fprintf(stderr, "wave_emu: misaligned %u-byte LDS store\n", size); abort(); }
    S.wlog[S.cur].push_back(w);
}
void lds_read(const void* p, void* out, unsigned size) {
    const uintptr_t a = (uintptr_t)p;
    if ((a & (size - 1u)) != 0) { fprintf(stderr, "wave_emu: misaligned %u-byte LDS load\n", size); abort(); }
    memcpy(out, p, size);
    uint8_t* o = (uint8_t*)out;
    for (const auto& w : S.wlog[S.cur]) {  // the lane's own earlier writes of this interval, in order
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
Idx tid() { return Idx{(unsigned)S.cur, 0, 0}; }
Idx bid() { return S.bid; }
Idx gdim() { return S.gdim; }
uint64_t clock64() { return S.clk += 7; }
void fail(const char* what, int tag) {
    fprintf(stderr, "wave_emu: %s (%d)\n", what, tag);
    abort();
}

const uint64_t* sync(uint64_t v, int tag, uint64_t* active) {
    const int me = S.cur;
    S.in[me] = v;
    S.tag[me] = tag;
    S.arrived_mask |= 1ull << me;
    S.arrived++;
    const unsigned g = S.gen;
    if (S.arrived == S.live) release();
    while (S.gen == g) swapcontext(&S.ctx[me], &S.sched);
    // a lane can be at most one rendezvous ahead of the slowest one, so generation g+1 is still intact
    *active = S.act[(g + 1u) & 1];
    return S.snap[(g + 1u) & 1];
}

// Runs `body` once per lane of one wavefront (workgroup of n_lanes <= 64 threads) as block `block` of `grid`.
void run_wave(const std::function<void()>& body, unsigned block, unsigned grid, int n_lanes) {
    if (!S.stacks_ready) {
        for (int i = 0; i < W; i++) S.stack[i] = (char*)malloc(STACK);
        S.stacks_ready = true;
    }
    S.body = body;
    S.bid = Idx{block, 0, 0};
    S.gdim = Idx{grid, 1, 1};
    S.n_lanes = n_lanes;
    S.live = n_lanes;
    S.arrived = 0;
    S.arrived_mask = 0;
    for (int i = 0; i < W; i++) S.done[i] = i >= n_lanes;
    for (int i = 0; i < n_lanes; i++) {
        getcontext(&S.ctx[i]);
        S.ctx[i].uc_stack.ss_sp = S.stack[i];
        S.ctx[i].uc_stack.ss_size = STACK;
        S.ctx[i].uc_link = &S.sched;
        makecontext(&S.ctx[i], (void (*)())trampoline, 0);
    }
    while (S.live > 0) {
        for (int i = 0; i < n_lanes; i++) {
            if (S.done[i]) continue;
            S.cur = i;
            swapcontext(&S.sched, &S.ctx[i]);
        }
    }
}
}  // namespace emu
