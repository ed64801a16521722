// f64_minmax_calib.hip - issue rate of the instructions of the 3DGRT hit buffer's insertion chain on gfx950 (MI355X), next to v_fma_f32:
// v_min_f64 / v_max_f64 (HitBufferT::insert: 32 per insertion), v_fma_f64, and the 32-bit alternatives (v_min_u32 / v_max_u32, v_cmp_lt_u64 +
// v_cndmask).  Same method as valu_calib.hip: 8 independent accumulators, W waves per SIMD on every SIMD, s_memtime inside the wave.
//   hipcc -O2 --offload-arch=gfx950 scripts/f64_minmax_calib.hip -o /tmp/f64_calib && /tmp/f64_calib
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
constexpr int kIters = 1024, kUnroll = 64;
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
enum Kind { FMA32 = 0, MIN64, MAX64, FMA64, MINU32, CMPU64, CNDMASK_S, NUM_KINDS };
static const char* kNames[NUM_KINDS] = {"v_fma_f32", "v_min_f64", "v_max_f64", "v_fma_f64", "v_min_u32", "v_cmp_lt_u64", "v_cndmask_b32(sgpr pair)"};
template <int KIND>
__global__ __launch_bounds__(256) void calib_kernel(double* __restrict__ out, unsigned long long* __restrict__ cycles, float seed) {
    float a[8];
    double d[8];
    uint32_t u[8];
    const float x = seed + threadIdx.x * 1e-7f, y = 0.999f;
    const double dx = x, dy = y;
    const uint32_t ux = threadIdx.x * 2654435761u;
    const unsigned long long mask = __builtin_amdgcn_readfirstlane((int)seed) ? 0x5555555555555555ull : 0x3333333333333333ull;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = x + i; d[i] = dx + i; u[i] = ux + i; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < kUnroll / 8; ++r) {
            if (KIND == FMA32) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                REP8(X)
#undef X
            } else if (KIND == MIN64) {
#define X(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dy));
                REP8(X)
#undef X
            } else if (KIND == MAX64) {
#define X(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dy));
                REP8(X)
#undef X
            } else if (KIND == FMA64) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(dx), "v"(dy));
                REP8(X)
#undef X
            } else if (KIND == MINU32) {
#define X(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ux));
                REP8(X)
#undef X
            } else if (KIND == CMPU64) {
#define X(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(d[i]), "v"(dy) : "vcc");
                REP8(X)
#undef X
            } else if (KIND == CNDMASK_S) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ux), "s"(mask));
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + d[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int KIND>
static void run(int w, double clock_ghz) {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * w;   // 256 threads = 4 waves = one per SIMD of a CU; w blocks per CU
    double* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 256 * 8); hipMalloc(&cyc, (size_t)blocks * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(calib_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(calib_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c((size_t)blocks * 4);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double n = (double)kIters * kUnroll;
    printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"memtime_ticks_per_inst_per_simd\": %.3f, \"wall_ms\": %.4f, \"cycles_per_inst_per_simd_at_max_clock\": %.3f}\n",
           kNames[KIND], w, (double)c[c.size() / 2] / (n * w), ms, ms * 1e6 * clock_ghz / (n * w));
    hipFree(out); hipFree(cyc);
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate * 1e-6;
    for (int w : {1, 4}) {
        run<FMA32>(w, ghz); run<MIN64>(w, ghz); run<MAX64>(w, ghz); run<FMA64>(w, ghz); run<MINU32>(w, ghz); run<CMPU64>(w, ghz);
    }
    return 0;
}
