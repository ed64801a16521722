// ref_block_emul.inl — runs a CUDA thread block on the host, one cooperative fiber per CUDA thread, so that the reference's
// tile-cooperative kernels (16x16 threads sharing a prefetch buffer, block barriers, 32-lane warp votes and shuffles) execute
// unmodified.  Everything here is emulation scaffolding written for this repository; it contains no reference code.
// TEST INFRASTRUCTURE ONLY (oracle/_ref, driven by tests/golden/make_golden.py in the build container).
//
// Model: one OS thread; the fibers of a block are resumed round-robin; a fiber that waits at a barrier yields.  `__shared__`
// variables become function-local statics (one block is resident at a time).  Warp = 32 consecutive linear thread ids, as on
// the hardware the reference targets.  A collective completes when every fiber of its group that has not yet returned from
// the kernel has arrived (threads that exited do not hold up a barrier).
#include <ucontext.h>
#include <cstdlib>
#include <functional>
#include <vector>

struct EmuDim3 { unsigned x = 1, y = 1, z = 1; };
static EmuDim3 blockIdx, blockDim, gridDim, threadIdx;
static constexpr int warpSize = 32;
#define __shared__ static
#define __global__
#define __launch_bounds__(...)

namespace emu {
struct Group {          // a barrier over `expected` fibers
    int expected = 0, arrived = 0, acc_and = 1;
    unsigned generation = 0;
    int result = 1;
    uint32_t slots[2][32];   // shuffle staging, double-buffered by the parity of the call count (one barrier per shuffle)
    void release() { result = acc_and; acc_and = 1; arrived = 0; ++generation; }
};
struct Fiber {
    ucontext_t ctx;
    std::vector<char> stack;
    bool done = false;
    EmuDim3 tid;
    int lane = 0, warp = 0;
    unsigned shuffle_count = 0;   // number of shuffles this fiber has issued (selects the staging buffer)
};
static std::vector<Fiber> g_fibers;
static std::vector<Group> g_warps;
static Group g_block;
static int g_current = -1;
static ucontext_t g_scheduler;
static std::function<void()> g_kernel;

static inline Fiber& self() { return g_fibers[g_current]; }
static inline void yield() { swapcontext(&self().ctx, &g_scheduler); }

static int arrive_and_wait(Group& g, int pred) {
    g.acc_and &= (pred != 0);
    const unsigned gen = g.generation;
    if (++g.arrived == g.expected) g.release();
    else while (g.generation == gen) yield();
    return g.result;
}
static void fiber_exit_from(Group& g) {   // a thread that returns from the kernel no longer takes part in barriers
    --g.expected;
    if (g.expected > 0 && g.arrived == g.expected) g.release();
}
static void fiber_main() {
    g_kernel();
    Fiber& f = self();
    f.done = true;
    fiber_exit_from(g_block);
    fiber_exit_from(g_warps[f.warp]);
    swapcontext(&f.ctx, &g_scheduler);
}

// runs `kernel` for every thread of one block (blockIdx / blockDim / gridDim set by the caller)
static void run_block(const std::function<void()>& kernel) {
    const int n = (int)(blockDim.x * blockDim.y * blockDim.z);
    const int n_warps = (n + warpSize - 1) / warpSize;
    g_kernel = kernel;
    g_fibers.resize(n);
    g_warps.assign(n_warps, Group());
    g_block = Group();
    g_block.expected = n;
    for (int i = 0; i < n; ++i) {
        Fiber& f = g_fibers[i];
        f.done = false;
        f.shuffle_count = 0;
        f.tid.x = i % blockDim.x; f.tid.y = (i / blockDim.x) % blockDim.y; f.tid.z = i / (blockDim.x * blockDim.y);
        f.lane = i % warpSize; f.warp = i / warpSize;
        g_warps[f.warp].expected++;
        if (f.stack.empty()) f.stack.resize(256 * 1024);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.data();
        f.ctx.uc_stack.ss_size = f.stack.size();
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, fiber_main, 0);
    }
    for (int live = n; live > 0;) {
        live = 0;
        for (int i = 0; i < n; ++i) {
            if (g_fibers[i].done) continue;
            g_current = i;
            threadIdx = g_fibers[i].tid;
            swapcontext(&g_scheduler, &g_fibers[i].ctx);
            if (!g_fibers[i].done) ++live;
        }
    }
    g_current = -1;
}

template <class T>
static T shuffle(T v, int src_lane_of_me(int lane, int arg), int arg) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    Fiber& f = self();
    Group& w = g_warps[f.warp];
    uint32_t* buf = w.slots[f.shuffle_count++ & 1u];
    std::memcpy(&buf[f.lane], &v, 4);
    arrive_and_wait(w, 1);
    const int src = src_lane_of_me(f.lane, arg);
    T r;
    std::memcpy(&r, &buf[(src >= 0 && src < warpSize) ? src : f.lane], 4);
    return r;
}
}  // namespace emu

static inline void __syncthreads() { emu::arrive_and_wait(emu::g_block, 1); }
static inline int __syncthreads_and(int pred) { return emu::arrive_and_wait(emu::g_block, pred); }
static inline int __all_sync(unsigned, int pred) { return emu::arrive_and_wait(emu::g_warps[emu::self().warp], pred); }
static inline unsigned __ballot_sync(unsigned, int pred) {   // one staging round: every lane posts its bit, then reads all 32
    const unsigned mine = pred ? 1u : 0u;
    emu::Fiber& f = emu::self();
    emu::Group& w = emu::g_warps[f.warp];
    uint32_t* buf = w.slots[f.shuffle_count++ & 1u];
    buf[f.lane] = mine;
    emu::arrive_and_wait(w, 1);
    unsigned m = 0;
    for (int l = 0; l < warpSize; ++l) m |= (buf[l] & 1u) << l;
    return m;
}
static inline int __ffs(unsigned v) { return v ? __builtin_ctz(v) + 1 : 0; }
template <class T> static inline T __shfl_xor_sync(unsigned, T v, int mask) {
    return emu::shuffle<T>(v, [](int lane, int m) { return lane ^ m; }, mask);
}
template <class T> static inline T __shfl_down_sync(unsigned, T v, unsigned delta) {
    return emu::shuffle<T>(v, [](int lane, int d) { return lane + d; }, (int)delta);
}
template <class T> static inline T __shfl_up_sync(unsigned, T v, unsigned delta) {
    return emu::shuffle<T>(v, [](int lane, int d) { return lane - d; }, (int)delta);
}
template <class T> static inline T __shfl_sync(unsigned, T v, int src) {
    return emu::shuffle<T>(v, [](int, int s) { return s; }, src);
}
