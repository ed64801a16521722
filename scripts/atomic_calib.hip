// atomic_calib.hip — throughput calibration of fp32 global_atomic_add on gfx950 (MI355X), the unit that binds the 3DGRT replay backward.
//
//   hipcc -O2 --offload-arch=gfx950 -munsafe-fp-atomics scripts/atomic_calib.hip -o /tmp/atomic_calib && /tmp/atomic_calib > profiles/rNN_atomic_calib.json
//
// The replay backward (csrc/grt_kernels.hip: grt_replay_bwd_kernel) turns every differentiated hit into one atomic instruction whose
// lanes 0..58 add consecutive float words of ONE particle's gradient rows (11 words of the [N,12] row + 48 words of the [N,48] row: five
// 64-byte lines), agent scope, no return value.  bench.py prices that kernel against the rate measured HERE, not against HBM bandwidth:
// the words never leave the L2 as individual transactions, so an "HBM fraction" of them says nothing.
//
// Patterns (every lane of every wave issues kIters atomic adds; 256 CUs x 8 waves x 4 SIMDs resident, all XCDs):
//   rows59      : the kernel's own shape - a wave adds 59 consecutive words of a pseudo-random row of a [R,64] table per instruction,
//                 R = 1 M rows (256 MB: far beyond the L2s), 5 lanes idle;
//   rows59_hot  : the same with R = 4096 rows (1 MB: every line lives in the L2 of its channel; many waves meet on a row);
//   line64      : 64 consecutive words of a random 256-byte-aligned block (full wave, four whole lines);
//   scatter     : every lane its own random word of the 256 MB table (64 lines per instruction: the reference's per-hit pattern);
//   same_word   : all waves on a handful of words (worst-case collision).
// Reported: float words per second, instructions per second, and words per clock per L2 channel (16 channels x 8 XCDs at the
// measured shader clock) for the first two.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                                      \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; }   \
    } while (0)

constexpr int kIters = 512;

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

enum Pattern { ROWS59 = 0, LINE64, SCATTER, SAME_WORD };

template <int PATTERN>
__global__ __launch_bounds__(64) void atomic_kernel(float* __restrict__ table, uint32_t rows_mask, uint32_t words_mask, float v) {
    const int lane = threadIdx.x;
    uint32_t s = blockIdx.x * 2654435761u + 12345u;         // wave-uniform stream of rows
    uint32_t sl = (blockIdx.x * 64u + lane) * 747796405u + 2891336453u;   // per-lane stream (scatter)
    for (int it = 0; it < kIters; ++it) {
        if (PATTERN == ROWS59) {
            const uint32_t row = (lcg(s) >> 8) & rows_mask;
            if (lane < 59) atomicAdd(table + (size_t)row * 64 + lane, v);
        } else if (PATTERN == LINE64) {
            const uint32_t row = (lcg(s) >> 8) & rows_mask;
            atomicAdd(table + (size_t)row * 64 + lane, v);
        } else if (PATTERN == SCATTER) {
            atomicAdd(table + ((lcg(sl) >> 4) & words_mask), v);
        } else {
            atomicAdd(table + (lane & 7), v);
        }
    }
}

template <int PATTERN>
static int run(const char* name, float* table, uint32_t rows, size_t words, int lanes_active, bool last) {
    const int blocks = 256 * 4 * 8 * 4;   // every SIMD of the chip holds its 8 waves for the length of the launch, four times over
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(atomic_kernel<PATTERN>, dim3(blocks), dim3(64), 0, 0, table, rows - 1, (uint32_t)(words - 1), 1e-6f);   // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(atomic_kernel<PATTERN>, dim3(blocks), dim3(64), 0, 0, table, rows - 1, (uint32_t)(words - 1), 1e-6f);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    const double instr = (double)blocks * kIters, wordsn = instr * lanes_active;
    printf("  \"%s\": {\"ms\": %.4f, \"atomic_instructions\": %.0f, \"words\": %.0f, \"words_per_s\": %.4e, \"instructions_per_s\": %.4e}%s\n", name, best,
           instr, wordsn, wordsn / (best * 1e-3), instr / (best * 1e-3), last ? "" : ",");
    return 0;
}

int main() {
    const size_t big_rows = 1u << 20, words = big_rows * 64;   // 256 MB
    float* table = nullptr;
    CHECK(hipMalloc(&table, words * sizeof(float)));
    CHECK(hipMemset(table, 0, words * sizeof(float)));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\n  \"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"iters_per_lane\": %d,\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000, kIters);
    printf("  \"note\": \"fp32 atomicAdd (global_atomic_add_f32, agent scope, no return), all XCDs saturated; rows59 = the 3DGRT replay backward's instruction shape\",\n");
    if (run<ROWS59>("rows59", table, (uint32_t)big_rows, words, 59, false)) return 1;
    if (run<ROWS59>("rows59_hot", table, 4096u, words, 59, false)) return 1;
    if (run<LINE64>("line64", table, (uint32_t)big_rows, words, 64, false)) return 1;
    if (run<SCATTER>("scatter", table, (uint32_t)big_rows, words, 64, false)) return 1;
    if (run<SAME_WORD>("same_word", table, 1u, words, 64, true)) return 1;
    printf("}\n");
    (void)hipFree(table);
    return 0;
}
