// valu_calib.hip — issue-rate calibration of wave64 fp32 VALU instructions on gfx950 (MI355X).
//
//   hipcc -O2 --offload-arch=gfx950 scripts/valu_calib.hip -o build/valu_calib && build/valu_calib > profiles/rNN_valu_calib.json
//
// bench.py's `valu` object prices a kernel's SQ_INSTS_VALU count in SIMD cycles; this program measures the cycles each
// instruction class really occupies the SIMD for, so that the constant is a measurement of this chip and not an assumption.
// For every instruction class a wave runs kIters x kUnroll instructions over 8 independent accumulators (no dependency
// stall: the same register is reused every 8 instructions) with W = 1, 2, 4 waves resident per SIMD on every SIMD of the chip
// (1024 x W waves). Cycles are read with s_memtime inside the wave (shader-clock ticks, MI355X_MICROARCH.md), so DVFS does
// not enter: cycles per instruction per SIMD = (wave's elapsed ticks) / (instructions per wave x W).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kIters = 2048;
constexpr int kUnroll = 64;  // instructions per loop body (8 accumulators x 8)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Kind { FMA = 0, PK_FMA, PK_MUL, PK_ADD, MUL, EXP, RCP, SQRT, DPP_ADD, CNDMASK, CMP, MIN3, NUM_KINDS };
static const char* kNames[NUM_KINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mul_f32", "v_exp_f32",
                                        "v_rcp_f32", "v_sqrt_f32", "v_add_f32_dpp(row_shr:1)", "v_cndmask_b32", "v_cmp_lt_f32",
                                        "v_min3_f32"};

template <int KIND>
__global__ __launch_bounds__(256) void calib_kernel(float* __restrict__ out, unsigned long long* __restrict__ cycles, float seed) {
    float a[8];
    v2f p[8];
    const float x = seed + threadIdx.x * 1e-7f, y = 0.999f;
    const v2f px = v2f{x, x}, py = v2f{y, y};
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = x + i; p[i] = v2f{x + i, x - i}; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < kUnroll / 8; ++r) {
            if (KIND == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                REP8(X)
#undef X
            } else if (KIND == PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(px), "v"(py));
                REP8(X)
#undef X
            } else if (KIND == PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(py));
                REP8(X)
#undef X
            } else if (KIND == PK_ADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(py));
                REP8(X)
#undef X
            } else if (KIND == MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
                REP8(X)
#undef X
            } else if (KIND == EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                REP8(X)
#undef X
            } else if (KIND == RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                REP8(X)
#undef X
            } else if (KIND == SQRT) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                REP8(X)
#undef X
            } else if (KIND == DPP_ADD) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(y));
                REP8(X)
#undef X
            } else if (KIND == CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(y) : );
                REP8(X)
#undef X
            } else if (KIND == CMP) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(y) : "vcc");
                REP8(X)
#undef X
            } else if (KIND == MIN3) {
#define X(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(int waves_per_simd, float* d_out, unsigned long long* d_cyc, std::vector<unsigned long long>& h_cyc, int num_cus,
                double& cyc_per_inst, double& wall_ms) {
    // one 256-thread workgroup = 4 waves = one wave per SIMD of a CU; W workgroups per CU are co-resident (8 VGPR-light waves fit)
    const int blocks = num_cus * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    calib_kernel<KIND><<<blocks, 256>>>(d_out, d_cyc, 1.0f);  // warm-up (clocks, code load)
    hipEventRecord(e0);
    calib_kernel<KIND><<<blocks, 256>>>(d_out, d_cyc, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    wall_ms = ms;
    hipMemcpy(h_cyc.data(), d_cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> c(h_cyc.begin(), h_cyc.begin() + blocks * 4);
    std::sort(c.begin(), c.end());
    const double med = (double)c[c.size() / 2];
    // s_memtime ticks at a fixed 100 MHz on gfx9 unless the shader clock is selected; report both raw ticks and the wall-clock view
    cyc_per_inst = med / ((double)kIters * kUnroll * waves_per_simd);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int num_cus = prop.multiProcessorCount;
    const double clock_ghz = prop.clockRate * 1e-6;  // kHz -> GHz (maximum shader clock)
    float* d_out;
    unsigned long long* d_cyc;
    const int max_blocks = num_cus * 8;
    hipMalloc(&d_out, sizeof(float) * max_blocks * 256);
    hipMalloc(&d_cyc, sizeof(unsigned long long) * max_blocks * 4);
    std::vector<unsigned long long> h(max_blocks * 4);
    printf("{\"device\": \"%s\", \"cus\": %d, \"max_clock_ghz\": %.3f, \"insts_per_wave\": %d, \"rows\": [\n", prop.gcnArchName, num_cus, clock_ghz,
           kIters * kUnroll);
    bool first = true;
    auto emit = [&](const char* name, int w, double ticks_per_inst, double ms) {
        // wall view: every SIMD issued W x kIters x kUnroll instructions in `ms`; at f GHz that is ms x 1e6 x f cycles
        const double wall_cyc_at_max = ms * 1e6 * clock_ghz / ((double)kIters * kUnroll * w);
        printf("%s  {\"inst\": \"%s\", \"waves_per_simd\": %d, \"memtime_ticks_per_inst_per_simd\": %.4f, \"wall_ms\": %.4f, "
               "\"cycles_per_inst_per_simd_at_max_clock\": %.3f}", first ? "" : ",\n", name, w, ticks_per_inst, ms, wall_cyc_at_max);
        first = false;
    };
    for (int w : {1, 2, 4}) {
        double c, ms;
#define RUN(K) run<K>(w, d_out, d_cyc, h, num_cus, c, ms); emit(kNames[K], w, c, ms);
        RUN(FMA) RUN(PK_FMA) RUN(PK_MUL) RUN(PK_ADD) RUN(MUL) RUN(EXP) RUN(RCP) RUN(SQRT) RUN(DPP_ADD) RUN(CNDMASK) RUN(CMP) RUN(MIN3)
#undef RUN
    }
    printf("\n]}\n");
    return 0;
}
