// common.hpp — shared host/device utilities of libgrut_amd (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/grut_amd.h"

#define GRUT_ABI_VERSION 5
#define GRUT_WAVE 64

namespace grut {

// ---- error plumbing -------------------------------------------------------------------------
void set_last_error(const char* fmt, ...);

#define GRUT_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            grut::set_last_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return GRUT_ERR_RUNTIME;                                                            \
        }                                                                                       \
    } while (0)

#define GRUT_CHECK(expr)                       \
    do {                                       \
        int _s = (expr);                       \
        if (_s != GRUT_OK) return _s;          \
    } while (0)

#define GRUT_REQUIRE(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            grut::set_last_error(__VA_ARGS__); \
            return GRUT_ERR_BAD_INPUT;         \
        }                                      \
    } while (0)

// ---- XCD-aware work assignment -----------------------------------------------------------------
// The dispatcher deals workgroups to the 8 XCDs round-robin by linear id, and every XCD has its own L2.  Mapping
// workgroup b of n to work item xcd_contiguous(b, n) gives each XCD ONE contiguous range of work items, so that the
// workgroups resident on an XCD at any time are neighbours (image tiles / pixel blocks that share particles and BVH
// nodes) instead of every 8th one of a range eight times as wide.  Bijective for any n.
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t b, uint32_t n) {
    const uint32_t xcd = b & 7u, idx = b >> 3, q = n >> 3, rem = n & 7u;
    return xcd * q + (xcd < rem ? xcd : rem) + idx;
}
#endif

// ---- grow-only device scratch (the role of CudaBuffer::enlarge, src/cudaBuffer.cpp:44-60) ---
// Where the scratch comes from: hipMalloc / hipFree, or the caller's allocator (grut_set_allocator, include/grut_amd.h — the Python
// plugins install torch's caching allocator, so a grown buffer's old block goes back to the pool the caller's tensors come from and
// no hipMalloc / hipFree — an implicit device synchronisation each — happens once the pool is warm).
struct ScratchAllocator {
    void* (*alloc)(void* user, uint64_t bytes) = nullptr;
    void (*release)(void* user, void* ptr) = nullptr;
    void* user = nullptr;
};
ScratchAllocator& scratch_allocator();   // (gut_api.hip)
// The stream of the API call in progress on this thread: a block that comes out of the caller's pool is zero-filled ON THAT STREAM
// (hipMemsetAsync), i.e. after whatever the block's previous owner still has queued there - torch's caching allocator hands a freed
// block straight back for reuse on the same stream - and before the first kernel of this call.  (A blocking hipMemset runs on the NULL
// stream, which torch's pooled non-blocking streams are not ordered against.)
inline hipStream_t& scratch_stream() {
    static thread_local hipStream_t s = nullptr;
    return s;
}
struct ScratchStreamScope {
    hipStream_t prev;
    explicit ScratchStreamScope(hipStream_t s) : prev(scratch_stream()) { scratch_stream() = s; }
    ~ScratchStreamScope() { scratch_stream() = prev; }
    ScratchStreamScope(const ScratchStreamScope&) = delete;
    ScratchStreamScope& operator=(const ScratchStreamScope&) = delete;
};
struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    bool external = false;   // ptr came from the caller's allocator
    int ensure(size_t need, float growth = 1.0f) {
        if (need <= bytes) return GRUT_OK;
        size_t n = (size_t)((double)need * growth);
        n = (n + 255) & ~(size_t)255;
        release();
        const ScratchAllocator& a = scratch_allocator();
        if (a.alloc) {
            ptr = a.alloc(a.user, n);
            if (!ptr) {
                set_last_error("scratch allocation of %zu bytes failed in the caller's allocator", n);
                return GRUT_ERR_RUNTIME;
            }
            external = true;
            // recycled memory of the caller's pool: start from zeros like the fresh pages of a first hipMalloc, filled on the stream
            // of the API call that asked (scratch_stream(): ordered against the block's previous use and this call's kernels)
            // (development: GRUT_POISON_SCRATCH=<byte> fills with that byte instead; GRUT_POISON_INDEX=<k> only the k-th allocation)
            {
                static int count = 0;
                const char* pz = getenv("GRUT_POISON_SCRATCH");
                const char* pi = getenv("GRUT_POISON_INDEX");
                const bool poison = pz && (!pi || atoi(pi) == count);
                if (pz && pi && atoi(pi) == count) fprintf(stderr, "[grut] poisoned allocation %d: %zu bytes\n", count, n);
                ++count;
                GRUT_HIP(hipMemsetAsync(ptr, poison ? atoi(pz) : 0, n, scratch_stream()));
            }
        } else {
            GRUT_HIP(hipMalloc(&ptr, n));
            external = false;
        }
        bytes = n;
        return GRUT_OK;
    }
    void release() {
        if (ptr) {
            const ScratchAllocator& a = scratch_allocator();
            if (external && a.release) a.release(a.user, ptr);
            else if (!external) (void)hipFree(ptr);
        }
        ptr = nullptr;
        bytes = 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(ptr); }
};

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ---- hipEvent ring for Tracer.timings (splatRaster.cpp:118-161) -----------------------------
struct EventTimer {
    static constexpr int kRing = 256;
    hipEvent_t start[kRing], stop[kRing];
    int count = 0, created = 0;
    int begin(hipStream_t s) {
        int i = count % kRing;
        if (i >= created) {
            GRUT_HIP(hipEventCreate(&start[i]));
            GRUT_HIP(hipEventCreate(&stop[i]));
            created = i + 1;
        }
        GRUT_HIP(hipEventRecord(start[i], s));
        return GRUT_OK;
    }
    int end(hipStream_t s) {
        int i = count % kRing;
        GRUT_HIP(hipEventRecord(stop[i], s));
        count++;
        return GRUT_OK;
    }
    // average ms since the last collect (-1 if none); synchronises on the last event
    float collect() {
        int n = count < kRing ? count : kRing;
        if (n == 0) return -1.f;
        double sum = 0;
        for (int i = 0; i < n; ++i) {
            float ms = 0;
            (void)hipEventSynchronize(stop[i]);
            (void)hipEventElapsedTime(&ms, start[i], stop[i]);
            sum += ms;
        }
        count = 0;
        return (float)(sum / n);
    }
    void destroy() {
        for (int i = 0; i < created; ++i) {
            (void)hipEventDestroy(start[i]);
            (void)hipEventDestroy(stop[i]);
        }
        created = 0;
    }
};

// ---- scan / sort primitives (scan_sort.hip) -------------------------------------------------
// inclusive scan of n u32; if gather != nullptr the input element i is in[gather[i]].
size_t scan_scratch_bytes(uint32_t n);
int inclusive_scan_u32(hipStream_t s, uint32_t n, const uint32_t* in, const uint32_t* gather, uint32_t* out,
                       void* scratch, size_t scratch_bytes);
// stable LSD radix sort on key bits [begin_bit, end_bit); ping-pongs between (keys,vals) and (keys_tmp,vals_tmp),
// *out_keys/*out_vals receive the buffers holding the sorted result.  `n_dev` (optional) points to the element
// count in device memory (<= n); blocks beyond it exit early, so `n` may be a capacity bound.
size_t sort_scratch_bytes(uint32_t n);
int sort_pairs_u32(hipStream_t s, uint32_t n, const uint32_t* n_dev, int begin_bit, int end_bit,
                   uint32_t* keys, uint32_t* vals, uint32_t* keys_tmp, uint32_t* vals_tmp,
                   void* scratch, size_t scratch_bytes, uint32_t** out_keys, uint32_t** out_vals, bool vals_iota = false);
// vals_iota: the payload of element i is i itself — `vals` is then never read (the first pass generates it) and need not be written
// by the producer of the keys; it is still used as one of the two ping-pong buffers.

}  // namespace grut

// ---- device math ----------------------------------------------------------------------------
#ifdef __HIPCC__
namespace grut {

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
// Separately rounded fp32 multiply / add.  The library is built with -ffp-contract=fast, under which the backend fuses a*b+c
// regardless of pragmas; quantities whose BITS matter (sort keys) go through these so that they never contract.
__device__ __forceinline__ float mul_rn(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float add_rn(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// the same with a wave-uniform first operand kept in a scalar register (no VGPR copy of the uniform)
__device__ __forceinline__ float mul_rn_u(float uniform, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(b)); return r; }
__device__ __forceinline__ float add_rn_u(float uniform, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(b)); return r; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// rows of a 3x3
struct m3 {
    f3 r0, r1, r2;
};
__device__ __forceinline__ f3 mul_rows(const m3& m, f3 p) { return f3{dot(m.r0, p), dot(m.r1, p), dot(m.r2, p)}; }
// m^T * g (matmul_bw_vec, mathUtils.cuh:451-456)
__device__ __forceinline__ f3 mul_cols(const m3& m, f3 g) {
    return f3{g.x * m.r0.x + g.y * m.r1.x + g.z * m.r2.x, g.x * m.r0.y + g.y * m.r1.y + g.z * m.r2.y,
              g.x * m.r0.z + g.y * m.r1.z + g.z * m.r2.z};
}
// rows of R^T from a (w,x,y,z) quaternion (models/gaussianParticles.cuh:39-59)
__device__ __forceinline__ m3 quat_wxyz_to_rotT(float r, float x, float y, float z) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
    const float rx = r * x, ry = r * y, rz = r * z;
    m3 m;
    m.r0 = f3{1.f - 2.f * (yy + zz), 2.f * (xy + rz), 2.f * (xz - ry)};
    m.r1 = f3{2.f * (xy - rz), 1.f - 2.f * (xx + zz), 2.f * (yz + rx)};
    m.r2 = f3{2.f * (xz + ry), 2.f * (yz - rx), 1.f - 2.f * (xx + yy)};
    return m;
}

// generalized Gaussian response (models/gaussianParticles.cuh:267-308); DEG is a compile-time constant
template <int DEG>
__device__ __forceinline__ float particle_response(float g) {
    if constexpr (DEG == 8) { const float g2 = g * g; return __expf(-0.000685871056241f * g2 * g2); }
    else if constexpr (DEG == 5) return __expf(-0.0185185185185f * g * g * sqrtf(g));
    else if constexpr (DEG == 4) return __expf(-0.0555555555556f * g * g);
    else if constexpr (DEG == 3) return __expf(-0.166666666667f * g * sqrtf(g));
    else if constexpr (DEG == 1) return __expf(-1.5f * sqrtf(g));
    else if constexpr (DEG == 0) return fmaxf(1.f - 0.329630334487f * sqrtf(g), 0.f);
    else return __expf(-0.5f * g);
}
// d response / d grayDist * upstream (models/gaussianParticles.cuh:223-265)
template <int DEG>
__device__ __forceinline__ float particle_response_grd(float g, float gres, float gresGrd) {
    if constexpr (DEG == 8) return (-0.000685871056241f * 4.f) * g * g * g * gres * gresGrd;
    else if constexpr (DEG == 5) return (-0.0185185185185f * 2.5f) * g * sqrtf(g) * gres * gresGrd;
    else if constexpr (DEG == 4) return (-0.0555555555556f * 2.f) * g * gres * gresGrd;
    else if constexpr (DEG == 3) return (-0.166666666667f * 1.5f) * sqrtf(g) * gres * gresGrd;
    else if constexpr (DEG == 1) return (-1.5f * 0.5f) * sqrtf(g) * gres * gresGrd;
    else if constexpr (DEG == 0) return gres > 0.f ? (0.5f * -0.329630334487f * rsqrtf(g)) * gresGrd : 0.f;
    else return -0.5f * gres * gresGrd;
}

// ---- wave64 helpers -------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {
    // v + dpp(v); lanes with no source (bound_ctrl) add 0
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(moved);
}
// full-wave sum; the total lands in lane 63 (LLVM AtomicOptimizer's DPP sequence for gfx9)
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0x111>(v);        // row_shr:1
    v = dpp_add<0x112>(v);        // row_shr:2
    v = dpp_add<0x114>(v);        // row_shr:4
    v = dpp_add<0x118>(v);        // row_shr:8
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
    return v;
}
// ---- wave64 reduce-scatter of 16 values ------------------------------------------------------
// In: every lane holds v[0..15].  Out: lane l returns the sum over ALL 64 lanes of v[l & 15].
// Four DPP butterfly steps inside each 16-lane row (row_mirror, row_half_mirror, quad xor 2, quad xor 1;
// each lane keeps the half of its values selected by one lane-id bit and adds its partner's copy of the
// same half: 8+4+2+1 DPP adds instead of 16 x 6), then two cross-row exchanges of the single survivor.
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_reduce_scatter16_rows(const float (&v)[16], int lane) {
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    float w[8], x[4], y[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float keep = b3 ? v[8 + i] : v[i], send = b3 ? v[i] : v[8 + i];
        w[i] = keep + dpp_perm<0x140>(send);  // row_mirror: partner = lane ^ 15
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b2 ? w[4 + i] : w[i], send = b2 ? w[i] : w[4 + i];
        x[i] = keep + dpp_perm<0x141>(send);  // row_half_mirror: partner = lane ^ 7
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b1 ? x[2 + i] : x[i], send = b1 ? x[i] : x[2 + i];
        y[i] = keep + dpp_perm<0x4E>(send);   // quad_perm [2,3,0,1]: partner = lane ^ 2
    }
    float z = (b0 ? y[1] : y[0]) + dpp_perm<0xB1>(b0 ? y[0] : y[1]);  // quad_perm [1,0,3,2]: partner = lane ^ 1
    return z;
}
// After the four in-row steps lane l holds the sum over its 16-lane ROW of v[l & 15]; the wave total of term t is the
// sum of lanes t, t+16, t+32, t+48.  wave_reduce_scatter16 finishes with two cross-row exchanges; the *_rows variant
// returns the per-row partials so that a consumer that goes through LDS anyway can add the four rows there.
__device__ __forceinline__ float wave_reduce_scatter16(const float (&v)[16], int lane) {
    float z = wave_reduce_scatter16_rows(v, lane);
    z += __shfl_xor(z, 16, 64);
    z += __shfl_xor(z, 32, 64);
    return z;
}

// gfx950's lane-swap instructions finish the reduction on the VALU (no LDS crossbar round trip): v_permlane32_swap exchanges the
// upper half of one register with the lower half of another, v_permlane16_swap the odd rows of one with the even rows of the
// other; applied to two copies of the same value, the sum of the pair is (lane + lane ^ 32) resp. (lane + lane ^ 16).
__device__ __forceinline__ float wave_reduce_scatter16_all(const float (&v)[16], int lane) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const float z = wave_reduce_scatter16_rows(v, lane);
    const v2u a = __builtin_amdgcn_permlane32_swap(__float_as_uint(z), __float_as_uint(z), false, false);
    const float s = __uint_as_float(a.x) + __uint_as_float(a.y);
    const v2u b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b.x) + __uint_as_float(b.y);   // every lane: wave total of term (lane & 15)
}

// ---- packed fp32: two pixels per lane ---------------------------------------------------------
// gfx950 issues one wave64 VALU instruction per 4 cycles per SIMD; v_pk_fma/mul/add_f32 do two floats per lane in
// that slot.  The compositing kernels therefore keep a PAIR of pixels per lane in float2 registers; per-particle
// (wave-uniform) operands are broadcast by the instruction's op_sel, so no shuffles are needed.
typedef float v2f __attribute__((ext_vector_type(2)));
struct p3 {
    v2f x, y, z;
};
__device__ __forceinline__ v2f splat(float s) { v2f r = {s, s}; return r; }
__device__ __forceinline__ v2f pfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pfma(float a, v2f b, v2f c) { return __builtin_elementwise_fma(splat(a), b, c); }
__device__ __forceinline__ v2f pdot(p3 a, p3 b) { return pfma(a.x, b.x, pfma(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ v2f pdot(f3 a, p3 b) { return pfma(a.x, b.x, pfma(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ p3 pcross(p3 a, p3 b) {
    return p3{pfma(a.y, b.z, -(a.z * b.y)), pfma(a.z, b.x, -(a.x * b.z)), pfma(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ v2f psel(bool c0, bool c1, v2f t, v2f f) { v2f r = {c0 ? t.x : f.x, c1 ? t.y : f.y}; return r; }
__device__ __forceinline__ v2f pmax0(v2f a) { v2f r = {fmaxf(a.x, 0.f), fmaxf(a.y, 0.f)}; return r; }
__device__ __forceinline__ v2f prcp(v2f a) { v2f r = {__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; return r; }

// largest grayDist g with particle_response<DEG>(g) > x  (x in (0,1); 0 when no g qualifies).  Lets the per-pixel
// accept test run on grayDist itself, without a transcendental: response > min_response && response*density > min_alpha
//   <=>  grayDist < response_gray_limit(max(min_response, min_alpha / density))
template <int DEG>
__device__ __forceinline__ float response_gray_limit(float x) {
    if (!(x < 1.f) || !(x > 0.f)) return x > 0.f ? 0.f : 3.0e38f;
    const float L = -logf(x);
    if constexpr (DEG == 8) return sqrtf(sqrtf(L / 0.000685871056241f));
    else if constexpr (DEG == 5) return powf(L / 0.0185185185185f, 0.4f);
    else if constexpr (DEG == 4) return sqrtf(L / 0.0555555555556f);
    else if constexpr (DEG == 3) return powf(L / 0.166666666667f, 2.f / 3.f);
    else if constexpr (DEG == 1) { const float t = L / 1.5f; return t * t; }
    else if constexpr (DEG == 0) { const float t = (1.f - x) / 0.329630334487f; return t * t; }
    else return 2.f * L;
}

__device__ __forceinline__ float wave_sum(float v) {
    v = wave_sum_to_lane63(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

}  // namespace grut
#endif  // __HIPCC__
