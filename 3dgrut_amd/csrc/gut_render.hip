// gut_render.hip — 3DGUT compositing for gfx950: front-to-back alpha compositing of the sorted tile lists and its
// gradient sweep.
//
// Reference behaviour restated (not translated): gutKBufferRenderer.cuh:199-352 (render, K = 0), :642-716
// (evalBackwardNoKBuffer, SH branch), common/rayPayload.cuh:75-193, rayPayloadBackward.cuh:30-73,
// models/gaussianParticles.cuh:350-422 (processHitFwd), :484-751 (processHitBwd).
//
// CDNA4 design.  Both sweeps are bound by fp32 VALU issue (one wave64 VALU instruction per 4 cycles per SIMD), not by
// HBM, so the kernels are organised around instruction count:
//   * one wave64 owns a 16x8 half tile and every lane carries a PAIR of pixels (rows y and y+4) in float2 registers,
//     so the multiply/add bulk of the per-(pixel, particle) math issues as v_pk_fma/mul/add_f32; per-particle operands
//     are wave-uniform LDS broadcasts selected by the packed instructions' op_sel;
//   * the accept test runs on the un-normalised quantities |v x u|^2 < g_max |v|^2 (u, v = ray origin / direction in
//     the particle's canonical frame, g_max = per-particle limit on grayDist derived from min_response and
//     min_alpha / density when the entry is staged): no transcendental on the reject path;
//   * when every ray of the wave shares its origin (pinhole / fisheye batches: rays_ori = 0 in camera space) the
//     canonical origin u is per particle and is staged with the record instead of being recomputed per pixel;
//   * gradient sweep: with a = u - v (v.u)/|v|^2 every geometric gradient is a multiple of a, so per hit a lane
//     produces  B = dL/d(R^T(o-mu))  (3),  M = B (x) (o - mu - t d)  (9, d rotT = M),  d density (1), d radiance (3)
//     = 16 terms; the two pixels are folded in-lane, summed over the wave through a transposition in LDS (lane l ends
//     up with a quarter of the wave total of term l mod 16) and written to the tile entry's own SLOT (one per entry and
//     half tile, no atomics: gradients are bitwise reproducible); quaternion, scale and position gradients are
//     contracted from (B, M) once per particle by the gather kernel, not per pixel.
#include <hip/hip_fp16.h>

#include "gut_internal.hpp"
// This file is compiled TWICE (3dgrut_amd/build.py): GRUT_RENDER_PART=0 -> gut_render.o, everything except the launchers of the sorted hit
// buffer with SH radiance, built with the backend's max-ILP scheduling strategy (the unsorted sweeps gain 1-2 %, the feature sweeps 2.5 %);
// GRUT_RENDER_PART=1 -> gut_render_k.o, launch_render_k_fwd / launch_render_k_bwd and the kernels they instantiate, with the default strategy
// (max-ILP costs the K = 16 frame 4 %: 10.87 -> 11.32 ms).  Undefined (a plain compile of the file): both parts, as before.
#ifndef GRUT_RENDER_PART
#define GRUT_RENDER_PART 2
#endif

// tuning switches of the gradient sweep (scripts/build_variant.sh -D...)
#ifndef GRUT_BWD_FULL_REDUCE
#define GRUT_BWD_FULL_REDUCE 2   // 0: row partials through LDS, 1: DPP butterfly + lane swaps, 2: transposition through LDS (r02j: 0.887 -> 0.85 ms)
#endif
#ifndef GRUT_BWD_WAVES
#define GRUT_BWD_WAVES 4   // waves per SIMD the register allocator is held to (0: its own choice, 130 VGPRs = 3 waves); measured r02b:
#endif                     // rows / 3 waves 0.917 ms, rows / 4 waves 0.910, lane-swap reduce / 3 waves 0.936, lane-swap reduce / 4 waves 0.891

namespace grut {

namespace {

// ---------------------------------------------------------------------------------------------
// rays (rayPayload.cuh:75-108, bounding_box.h:89-140)
// ---------------------------------------------------------------------------------------------
struct Ray {
    f3 o, d;
    float tmin, tmax;
    bool valid;
    bool inside;   // pixel lies in the image (a ray that misses the scene box is inside but not valid)
};
__device__ __forceinline__ void swapf(float& a, float& b) { const float t = a; a = b; b = t; }
// (a.x b.x + a.y b.y) + a.z b.z with every product and sum rounded on its own (the checker's v3_dot under -ffp-contract=off)
__device__ __forceinline__ float rn_dot3(float ax, float ay, float az, f3 b) { return add_rn(add_rn(mul_rn(ax, b.x), mul_rn(ay, b.y)), mul_rn(az, b.z)); }
__device__ __forceinline__ float rn_dot3(f3 a, f3 b) { return rn_dot3(a.x, a.y, a.z, b); }
__device__ __forceinline__ Ray init_ray(const GutParams& P, const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                        int px, int py) {
    Ray r;
    r.valid = false;
    r.inside = false;
    r.o = mk3(0.f, 0.f, 0.f);
    r.d = mk3(0.f, 0.f, 1.f);
    r.tmin = r.tmax = 0.f;
    if (px >= P.W || py >= P.H) return r;
    r.inside = true;
    const size_t pix = (size_t)py * P.W + px;
    const f3 so = mk3(ray_o[3 * pix], ray_o[3 * pix + 1], ray_o[3 * pix + 2]);
    const f3 sd = mk3(ray_d[3 * pix], ray_d[3 * pix + 1], ray_d[3 * pix + 2]);
    const FramePoses& FP = frame_poses(P);
    const float* R = FP.s2w_R;
    // world-space ray = pose . (o, d) (rayPayload.cuh:75-108) in the checker's operation order - rows dotted left to right, separately
    // rounded, then the translation: the sorted mode orders hits by distances computed from these bits (oracle_order_hit_t)
    r.o = mk3(add_rn(rn_dot3(R[0], R[1], R[2], so), FP.s2w_t[0]), add_rn(rn_dot3(R[3], R[4], R[5], so), FP.s2w_t[1]),
              add_rn(rn_dot3(R[6], R[7], R[8], so), FP.s2w_t[2]));
    r.d = mk3(rn_dot3(R[0], R[1], R[2], sd), rn_dot3(R[3], R[4], R[5], sd), rn_dot3(R[6], R[7], R[8], sd));
    const float lo = -1e6f, hi = 1e6f, big = 3.4028234663852886e+38f;
    float tmin = (lo - r.o.x) / r.d.x, tmax = (hi - r.o.x) / r.d.x;
    if (tmin > tmax) swapf(tmin, tmax);
    float tymin = (lo - r.o.y) / r.d.y, tymax = (hi - r.o.y) / r.d.y;
    if (tymin > tymax) swapf(tymin, tymax);
    bool miss = (tmin > tymax) || (tymin > tmax);
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (lo - r.o.z) / r.d.z, tzmax = (hi - r.o.z) / r.d.z;
    if (tzmin > tzmax) swapf(tzmin, tzmax);
    miss = miss || (tmin > tzmax) || (tzmin > tmax);
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    if (miss) { tmin = big; tmax = big; }
    r.tmin = fmaxf(tmin, 0.f);
    r.tmax = tmax;
    r.valid = r.tmax > r.tmin;
    return r;
}

// The pixel pair of a lane and the wave-level facts about its rays.
struct RayPair {
    p3 o, d;
    v2f tmin, tmax;
    bool valid0, valid1;
    bool inside0, inside1;
    int px, py0, py1;
    bool uniform_origin;  // every valid ray of the wave starts at `origin`
    f3 origin;
};
__device__ __forceinline__ RayPair init_ray_pair(const GutParams& P, const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                 uint32_t tile, uint32_t half, int lane) {
    RayPair rp;
    rp.px = (int)(tile % P.gx) * 16 + (lane & 15);
    rp.py0 = (int)(tile / P.gx) * 16 + (int)half * 8 + (lane >> 4);
    rp.py1 = rp.py0 + 4;
    const Ray a = init_ray(P, ray_o, ray_d, rp.px, rp.py0), b = init_ray(P, ray_o, ray_d, rp.px, rp.py1);
    rp.o = p3{v2f{a.o.x, b.o.x}, v2f{a.o.y, b.o.y}, v2f{a.o.z, b.o.z}};
    rp.d = p3{v2f{a.d.x, b.d.x}, v2f{a.d.y, b.d.y}, v2f{a.d.z, b.d.z}};
    rp.tmin = v2f{a.tmin, b.tmin};
    rp.tmax = v2f{a.tmax, b.tmax};
    rp.valid0 = a.valid;
    rp.valid1 = b.valid;
    rp.inside0 = a.inside;
    rp.inside1 = b.inside;
    // wave-uniform origin?  take the first valid ray as the candidate
    const unsigned long long m0 = __ballot(a.valid), m1 = __ballot(b.valid);
    f3 cand = mk3(0.f, 0.f, 0.f);
    if (m0) {
        const int src = __ffsll((long long)m0) - 1;
        cand = mk3(__shfl(a.o.x, src, 64), __shfl(a.o.y, src, 64), __shfl(a.o.z, src, 64));
    } else if (m1) {
        const int src = __ffsll((long long)m1) - 1;
        cand = mk3(__shfl(b.o.x, src, 64), __shfl(b.o.y, src, 64), __shfl(b.o.z, src, 64));
    }
    const bool same0 = !a.valid || (a.o.x == cand.x && a.o.y == cand.y && a.o.z == cand.z);
    const bool same1 = !b.valid || (b.o.x == cand.x && b.o.y == cand.y && b.o.z == cand.z);
    rp.uniform_origin = __all(same0 && same1);
    rp.origin = cand;
    return rp;
}

// block -> (virtual tile, half) with both halves of a tile on one XCD (block b runs on XCD b % 8)
// i -> (i * k') mod n with k' the first number >= k coprime to n: a bijection of [0, n) that sends neighbours far apart.  Launch
// order = memory order keeps the waves in flight on one band of the image, all heavy or all light at a time; a strided order mixes them.
__device__ __forceinline__ uint32_t stride_permute(uint32_t i, uint32_t n, uint32_t k) {
    if (n < 3u) return i;
    k %= n;
    if (k < 2u) k = 2u;
    while (true) {
        uint32_t a = k, b = n;
        while (b) { const uint32_t t = a % b; a = b; b = t; }
        if (a == 1u) break;
        ++k;
    }
    return (uint32_t)(((unsigned long long)i * k) % n);
}
__device__ __forceinline__ void half_mapping(uint32_t b, uint32_t& vtile, uint32_t& half) {
    const uint32_t xcd = b & 7u, slot = b >> 3;
    vtile = ((slot >> 1) << 3) + xcd;
    half = slot & 1u;
}

// ---------------------------------------------------------------------------------------------
// staged tile entry: 6 x float4 in LDS
//   r0 = M.r0, pos.x | r1 = M.r1, pos.y | r2 = M.r2, pos.z      M = diag(1/scale) R^T  (gaussianParticles.slang:96-110)
//   r3 = scale.xyz (fwd) or 1/scale.xyz (bwd), density
//   r4 = clamped radiance rgb, g_max
//   r5 = u0 = M (origin - pos) (uniform-origin waves), as_float(expansion position of the entry)
// Padding entries (index 0xFFFFFFFF) get M = I and g_max = 0: finite everywhere, never accepted.
// ---------------------------------------------------------------------------------------------
constexpr int kRecQuads = 6;

struct RawEntry {
    uint32_t idx, pos;   // particle, expansion position of the entry (its gradient slot)
    float4 a, q, s;
    f3 rgb;
};
// the sorted lists carry expansion positions; pos_particle maps them to particles (0xFFFFFFFF = padding)
// direct lists (rec64 != null, GutParams::rec64): the sorted payloads ARE the particles, one 64-byte record each; the gradient slot
// of an entry is the particle's part_offset (in the record) + the ordinal in the upper key bits
struct EntryLists {
    const uint32_t* __restrict__ sorted_pos;
    const uint32_t* __restrict__ pos_particle;
    const float4* __restrict__ rec64;
    const uint32_t* __restrict__ sorted_keys;
    uint32_t ord_shift;
};
__host__ __device__ inline EntryLists entry_lists(const GutParams& P, const uint32_t* sorted_pos, const uint32_t* pos_particle) {
    return EntryLists{sorted_pos, P.rec64 ? nullptr : pos_particle, P.rec64, P.sorted_keys, P.ord_shift};
}
template <bool WITH_POS = true>
__device__ __forceinline__ RawEntry load_entry(uint32_t e, uint32_t end, const EntryLists& lists,
                                               const float4* __restrict__ density12, const float* __restrict__ rgb) {
    RawEntry r;
    r.idx = 0xFFFFFFFFu;
    r.pos = 0u;
    r.a = r.q = r.s = make_float4(0.f, 0.f, 0.f, 0.f);
    r.rgb = mk3(0.f, 0.f, 0.f);
    if (e < end && lists.rec64) {
        r.idx = lists.sorted_pos[e];
        if (r.idx != 0xFFFFFFFFu) {
            const float4* rec = lists.rec64 + 4 * (size_t)r.idx;
            r.a = rec[0]; r.q = rec[1]; r.s = rec[2];
            const float4 c = rec[3];
            r.rgb = mk3(c.x, c.y, c.z);
            if (WITH_POS) r.pos = __float_as_uint(r.s.w) + (lists.sorted_keys[e] >> lists.ord_shift);
        }
    } else if (e < end) {
        r.pos = lists.sorted_pos[e];
        r.idx = lists.pos_particle[r.pos];
        if (r.idx != 0xFFFFFFFFu) {
            r.a = density12[3 * (size_t)r.idx + 0];
            r.q = density12[3 * (size_t)r.idx + 1];
            r.s = density12[3 * (size_t)r.idx + 2];
            if (rgb) r.rgb = mk3(rgb[3 * (size_t)r.idx], rgb[3 * (size_t)r.idx + 1], rgb[3 * (size_t)r.idx + 2]);
        }
    }
    return r;
}
template <int DEG, bool INVERSE_SCALE>
__device__ __forceinline__ void stage_entry(const GutParams& P, const RawEntry& r, bool uniform_origin, f3 origin, float4* __restrict__ rec) {
    float4 r0 = make_float4(1.f, 0.f, 0.f, 0.f), r1 = make_float4(0.f, 1.f, 0.f, 0.f), r2 = make_float4(0.f, 0.f, 1.f, 0.f);
    float4 r3 = make_float4(1.f, 1.f, 1.f, 0.f), r4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 r5 = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
    if (r.idx != 0xFFFFFFFFu) {
        const m3 rt = quat_wxyz_to_rotT(r.q.x, r.q.y, r.q.z, r.q.w);
        const float ix = __builtin_amdgcn_rcpf(r.s.x), iy = __builtin_amdgcn_rcpf(r.s.y), iz = __builtin_amdgcn_rcpf(r.s.z);
        r0 = make_float4(rt.r0.x * ix, rt.r0.y * ix, rt.r0.z * ix, r.a.x);
        r1 = make_float4(rt.r1.x * iy, rt.r1.y * iy, rt.r1.z * iy, r.a.y);
        r2 = make_float4(rt.r2.x * iz, rt.r2.y * iz, rt.r2.z * iz, r.a.z);
        r3 = INVERSE_SCALE ? make_float4(ix, iy, iz, r.a.w) : make_float4(r.s.x, r.s.y, r.s.z, r.a.w);
        const float need = fmaxf(P.min_response, P.min_alpha / r.a.w);   // response must exceed this
        const float gmax = (P.max_alpha > P.min_alpha && r.a.w > 0.f) ? response_gray_limit<DEG>(need) : 0.f;
        r4 = make_float4(fmaxf(r.rgb.x, 0.f), fmaxf(r.rgb.y, 0.f), fmaxf(r.rgb.z, 0.f), gmax);
        if (uniform_origin) {
            const f3 dl = origin - mk3(r.a.x, r.a.y, r.a.z);
            r5.x = dot(mk3(r0.x, r0.y, r0.z), dl);
            r5.y = dot(mk3(r1.x, r1.y, r1.z), dl);
            r5.z = dot(mk3(r2.x, r2.y, r2.z), dl);
        }
        r5.w = __uint_as_float(r.pos);
    }
    rec[0] = r0; rec[1] = r1; rec[2] = r2; rec[3] = r3; rec[4] = r4; rec[5] = r5;
}

// ---------------------------------------------------------------------------------------------
// Half-tile culling of staged entries (round 4).  A tile's list is binned for the 16x16 tile by the reference's 2-D test; the wave owns
// 16x8 of it, and 36 % of the entries a wave evaluates on the bench frame are accepted by none of its 128 pixels - each
// of them costs the pair geometry.  The accept test is a LINE - sphere test in the particle's canonical frame (|v x u|^2 < gmax |v|^2:
// the line through u along v passes the origin closer than r = sqrt(gmax)), so an entry whose sphere misses every line of the wave can
// be dropped when it is staged, and no result changes.  The lines of a wave are bounded, for ANY camera model, by the pyramid of its
// own rays: in a frame (a, e1, e2) around the first valid ray every ray is a + x e1 + y e2 up to scale, and the wave reduces
// [x0, x1] x [y0, y1] once.  M maps the pyramid's four faces to planes through u spanned by (M c, M e2) / (M c, M e1) with c a corner
// direction; det[M c, M e1, M e2] = det M > 0 fixes which side is out.  The lines also extend BEHIND the apex; that mirror pyramid lies
// behind the plane through u with normal M e1 x M e2 (the image of the camera plane normal to a), so the entry is dropped only if its
// sphere is wholly in front of that plane and wholly outside one face.  All comparisons carry r' = 1.001 r + 1e-5 |u| (the accept test
// and these triple products both cancel to ~1e-7 |u|).
// ---------------------------------------------------------------------------------------------
struct WavePyramid {
    bool on;          // wave-uniform: the bound exists (>= 1 valid ray, every ray within 60 degrees of the first)
    f3 c00, e1, e2;   // corner direction a + x0 e1 + y0 e2 and the frame's tangents
    float dx, dy;     // x1 - x0, y1 - y0
};
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = fminf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float uniform(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
// (up to two rays per lane: the pixel pair of the half-tile waves, one ray of the quarter-tile waves with valid1 = false)
__device__ __forceinline__ WavePyramid wave_pyramid(bool valid0, f3 d0, bool valid1, f3 d1) {
    WavePyramid w;
    w.on = false;
    w.c00 = w.e1 = w.e2 = mk3(0.f, 0.f, 0.f);
    w.dx = w.dy = 0.f;
    const unsigned long long m0 = __ballot(valid0), m1 = __ballot(valid1);
    if (!(m0 | m1)) return w;
    f3 a;
    if (m0) { const int src = __ffsll((long long)m0) - 1; a = mk3(__shfl(d0.x, src, 64), __shfl(d0.y, src, 64), __shfl(d0.z, src, 64)); }
    else { const int src = __ffsll((long long)m1) - 1; a = mk3(__shfl(d1.x, src, 64), __shfl(d1.y, src, 64), __shfl(d1.z, src, 64)); }
    a = a * __builtin_amdgcn_rsqf(dot(a, a));
    a = mk3(uniform(a.x), uniform(a.y), uniform(a.z));
    // any unit tangent pair: e1 = a x (the axis a is least aligned with), e2 = a x e1
    const float ax = fabsf(a.x), ay = fabsf(a.y), az = fabsf(a.z);
    const f3 k = (ax <= ay && ax <= az) ? mk3(1.f, 0.f, 0.f) : (ay <= az ? mk3(0.f, 1.f, 0.f) : mk3(0.f, 0.f, 1.f));
    f3 e1 = cross(a, k);
    e1 = e1 * __builtin_amdgcn_rsqf(dot(e1, e1));
    const f3 e2 = cross(a, e1);
    const float q0 = dot(d0, a), q1 = dot(d1, a);
    const float l0 = __builtin_amdgcn_sqrtf(dot(d0, d0)), l1 = __builtin_amdgcn_sqrtf(dot(d1, d1));
    const bool ok0 = !valid0 || q0 > 0.5f * l0, ok1 = !valid1 || q1 > 0.5f * l1;
    if (!__all(ok0 && ok1)) return w;
    const float big = 3.0e38f;
    const float i0 = 1.f / q0, i1 = 1.f / q1;
    const float x_0 = dot(d0, e1) * i0, y_0 = dot(d0, e2) * i0, x_1 = dot(d1, e1) * i1, y_1 = dot(d1, e2) * i1;
    const float xlo = wave_min(fminf(valid0 ? x_0 : big, valid1 ? x_1 : big)), xhi = -wave_min(fminf(valid0 ? -x_0 : big, valid1 ? -x_1 : big));
    const float ylo = wave_min(fminf(valid0 ? y_0 : big, valid1 ? y_1 : big)), yhi = -wave_min(fminf(valid0 ? -y_0 : big, valid1 ? -y_1 : big));
    const float pad = 1e-6f;   // tangent units (a pixel is ~1e-3): the rounding of x, y themselves
    const float x0 = uniform(xlo) - pad, x1 = uniform(xhi) + pad, y0 = uniform(ylo) - pad, y1 = uniform(yhi) + pad;
    w.on = true;
    w.c00 = a + e1 * x0 + e2 * y0;
    w.e1 = mk3(uniform(e1.x), uniform(e1.y), uniform(e1.z));
    w.e2 = mk3(uniform(e2.x), uniform(e2.y), uniform(e2.z));
    w.c00 = mk3(uniform(w.c00.x), uniform(w.c00.y), uniform(w.c00.z));
    w.dx = x1 - x0;
    w.dy = y1 - y0;
    return w;
}
__device__ __forceinline__ WavePyramid wave_pyramid(const RayPair& rp) {
    return wave_pyramid(rp.valid0, mk3(rp.d.x.x, rp.d.y.x, rp.d.z.x), rp.valid1, mk3(rp.d.x.y, rp.d.y.y, rp.d.z.y));
}
// does the staged record's sphere miss every line of the wave?  (rec as stage_entry builds it, uniform-origin form: r5.xyz = u)
__device__ __forceinline__ bool pyramid_misses(const WavePyramid& w, float4 r0, float4 r1, float4 r2, float gmax, float4 r5) {
    const f3 m0 = mk3(r0.x, r0.y, r0.z), m1 = mk3(r1.x, r1.y, r1.z), m2 = mk3(r2.x, r2.y, r2.z);
    const f3 u = mk3(r5.x, r5.y, r5.z);
    const f3 v00 = mk3(dot(m0, w.c00), dot(m1, w.c00), dot(m2, w.c00));
    const f3 f1 = mk3(dot(m0, w.e1), dot(m1, w.e1), dot(m2, w.e1)), f2 = mk3(dot(m0, w.e2), dot(m1, w.e2), dot(m2, w.e2));
    const f3 v10 = v00 + f1 * w.dx, v01 = v00 + f2 * w.dy;
    const float rr = 1.001f * __builtin_amdgcn_sqrtf(gmax) + 1e-5f * __builtin_amdgcn_sqrtf(dot(u, u));
    const float r2lim = rr * rr;
    // signed distance of the sphere's centre (the origin) OUT of the face, times |n|:  s = -(u . n_out)
    const f3 nx0 = cross(v00, f2), nx1 = cross(v10, f2), ny0 = cross(v00, f1), ny1 = cross(v01, f1), wf = cross(f1, f2);
    const float sx0 = -dot(u, nx0), sx1 = dot(u, nx1), sy0 = dot(u, ny0), sy1 = -dot(u, ny1), sf = -dot(u, wf);
    const bool front = sf > 0.f && sf * sf > r2lim * dot(wf, wf);
    const bool out = (sx0 > 0.f && sx0 * sx0 > r2lim * dot(nx0, nx0)) || (sx1 > 0.f && sx1 * sx1 > r2lim * dot(nx1, nx1)) ||
                     (sy0 > 0.f && sy0 * sy0 > r2lim * dot(ny0, ny0)) || (sy1 > 0.f && sy1 * sy1 > r2lim * dot(ny1, ny1));
    return front && out;
}

// canonical-frame ray of the pixel pair against one staged entry, and the accept test
struct PairGeom {
    p3 u, v;        // canonical origin, canonical (un-normalised) direction
    v2f l2, cc;     // |v|^2, |v x u|^2  (grayDist = cc / l2)
    bool acc0, acc1;
};
template <bool UNI>
__device__ __forceinline__ PairGeom pair_geometry(const RayPair& rp, const float4* __restrict__ rec) {
    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
    PairGeom g;
    const f3 m0 = mk3(r0.x, r0.y, r0.z), m1 = mk3(r1.x, r1.y, r1.z), m2 = mk3(r2.x, r2.y, r2.z);
    g.v = p3{pdot(m0, rp.d), pdot(m1, rp.d), pdot(m2, rp.d)};
    if (UNI) {
        const float4 r5 = rec[5];
        g.u = p3{splat(r5.x), splat(r5.y), splat(r5.z)};
    } else {
        const p3 dl = p3{rp.o.x - r0.w, rp.o.y - r1.w, rp.o.z - r2.w};
        g.u = p3{pdot(m0, dl), pdot(m1, dl), pdot(m2, dl)};
    }
    g.l2 = pdot(g.v, g.v);
    const p3 c = pcross(g.v, g.u);
    g.cc = pdot(c, c);
    const v2f lim = rec[4].w * g.l2;
    g.acc0 = g.cc.x < lim.x;
    g.acc1 = g.cc.y < lim.y;
    return g;
}


// ---------------------------------------------------------------------------------------------
// Experiment (round 5, -DGRUT_FWD_MFMA=1): the canonical direction v = M d of the pair geometry on the matrix pipe.
// v_mfma_f32_4x4x1_16b_f32 multiplies, in each of 16 blocks of four lanes, a 4 x 1 column (A: one value per lane, row = lane & 3) with a
// 1 x 4 row (B: one value per lane, column = lane & 3) and adds the 4 x 4 product to the accumulator (register r of a lane = row r of its
// column).  With A = a coefficient of M of staged entry (group + (lane & 3)) and B = the lane's own ray direction component, register r
// of every lane receives coefficient(entry group + r) x d(own pixel): three chained instructions (z, y, x - the order of pdot) give one
// component of v for FOUR entries; fp32 MFMA is an exact fmaf chain on gfx950, so the bits are pdot's.  18 instructions per group of four
// entries and pixel pair replace 4 x 9 packed multiply-adds on the VALU pipe; the accumulators of the next group are issued before the
// current group is composited.
// ---------------------------------------------------------------------------------------------
#ifndef GRUT_FWD_MFMA
#define GRUT_FWD_MFMA 0
#endif
typedef float v4f __attribute__((ext_vector_type(4)));
struct GroupV {
    v4f x0, x1, y0, y1, z0, z1;   // component of v, pixel of the pair; register r = entry group + r
};
__device__ __forceinline__ GroupV group_directions(const RayPair& rp, const float4* __restrict__ s_rec, int jg, int lane) {
    const float4* rec = &s_rec[(jg + (lane & 3)) * kRecQuads];
    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    GroupV g;
    g.x0 = __builtin_amdgcn_mfma_f32_4x4x1f32(r0.z, rp.d.z.x, zero, 0, 0, 0);
    g.x1 = __builtin_amdgcn_mfma_f32_4x4x1f32(r0.z, rp.d.z.y, zero, 0, 0, 0);
    g.y0 = __builtin_amdgcn_mfma_f32_4x4x1f32(r1.z, rp.d.z.x, zero, 0, 0, 0);
    g.y1 = __builtin_amdgcn_mfma_f32_4x4x1f32(r1.z, rp.d.z.y, zero, 0, 0, 0);
    g.z0 = __builtin_amdgcn_mfma_f32_4x4x1f32(r2.z, rp.d.z.x, zero, 0, 0, 0);
    g.z1 = __builtin_amdgcn_mfma_f32_4x4x1f32(r2.z, rp.d.z.y, zero, 0, 0, 0);
    g.x0 = __builtin_amdgcn_mfma_f32_4x4x1f32(r0.y, rp.d.y.x, g.x0, 0, 0, 0);
    g.x1 = __builtin_amdgcn_mfma_f32_4x4x1f32(r0.y, rp.d.y.y, g.x1, 0, 0, 0);
    g.y0 = __builtin_amdgcn_mfma_f32_4x4x1f32(r1.y, rp.d.y.x, g.y0, 0, 0, 0);
    g.y1 = __builtin_amdgcn_mfma_f32_4x4x1f32(r1.y, rp.d.y.y, g.y1, 0, 0, 0);
    g.z0 = __builtin_amdgcn_mfma_f32_4x4x1f32(r2.y, rp.d.y.x, g.z0, 0, 0, 0);
    g.z1 = __builtin_amdgcn_mfma_f32_4x4x1f32(r2.y, rp.d.y.y, g.z1, 0, 0, 0);
    g.x0 = __builtin_amdgcn_mfma_f32_4x4x1f32(r0.x, rp.d.x.x, g.x0, 0, 0, 0);
    g.x1 = __builtin_amdgcn_mfma_f32_4x4x1f32(r0.x, rp.d.x.y, g.x1, 0, 0, 0);
    g.y0 = __builtin_amdgcn_mfma_f32_4x4x1f32(r1.x, rp.d.x.x, g.y0, 0, 0, 0);
    g.y1 = __builtin_amdgcn_mfma_f32_4x4x1f32(r1.x, rp.d.x.y, g.y1, 0, 0, 0);
    g.z0 = __builtin_amdgcn_mfma_f32_4x4x1f32(r2.x, rp.d.x.x, g.z0, 0, 0, 0);
    g.z1 = __builtin_amdgcn_mfma_f32_4x4x1f32(r2.x, rp.d.x.y, g.z1, 0, 0, 0);
    return g;
}
// the rest of pair_geometry<true> for a direction that is already there
__device__ __forceinline__ PairGeom pair_geometry_given_v(p3 v, const float4* __restrict__ rec) {
    PairGeom g;
    g.v = v;
    const float4 r5 = rec[5];
    g.u = p3{splat(r5.x), splat(r5.y), splat(r5.z)};
    g.l2 = pdot(g.v, g.v);
    const p3 c = pcross(g.v, g.u);
    g.cc = pdot(c, c);
    const v2f lim = rec[4].w * g.l2;
    g.acc0 = g.cc.x < lim.x;
    g.acc1 = g.cc.y < lim.y;
    return g;
}

// particle_response<DEG> for a pixel pair (one v_exp_f32 per pixel, the polynomial part packed)
template <int DEG>
__device__ __forceinline__ v2f pair_response(v2f g) {
    constexpr float kLog2e = 1.4426950408889634f;
    if constexpr (DEG == 2) {
        const v2f e = g * (-0.5f * kLog2e);
        return v2f{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    } else if constexpr (DEG == 4) {
        const v2f e = (g * g) * (-0.0555555555556f * kLog2e);
        return v2f{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    } else {
        return v2f{particle_response<DEG>(g.x), particle_response<DEG>(g.y)};
    }
}

// the [H,W,4] image: fp32, or IEEE half with FEATURE_OUTPUT_HALF (rayPayload.cuh:176-186 writes __float2half of every component;
// rayPayloadBackward.cuh:50-58 reads them back)
__device__ __forceinline__ void store_fd(const GutParams& P, float4* __restrict__ out_fd, size_t pix, float4 o) {
    if (P.out_half) {
        __half* h = reinterpret_cast<__half*>(out_fd) + 4 * pix;
        h[0] = __float2half(o.x); h[1] = __float2half(o.y); h[2] = __float2half(o.z); h[3] = __float2half(o.w);
    } else {
        out_fd[pix] = o;
    }
}
__device__ __forceinline__ float4 load_fd(const GutParams& P, const float4* __restrict__ fd, size_t pix) {
    if (P.out_half) {
        const __half* h = reinterpret_cast<const __half*>(fd) + 4 * pix;
        return make_float4(__half2float(h[0]), __half2float(h[1]), __half2float(h[2]), __half2float(h[3]));
    }
    return fd[pix];
}
// the plugin's `pred_features` / `pred_opacity` as contiguous tensors of their own (GutFrame::out_features / out_opacity)
__device__ __forceinline__ void write_split_outputs(const GutParams& P, size_t pix, float4 o) {
    if (P.out_features) { P.out_features[3 * pix] = o.x; P.out_features[3 * pix + 1] = o.y; P.out_features[3 * pix + 2] = o.z; }
    if (P.out_opacity) P.out_opacity[pix] = o.w;
}

// ---------------------------------------------------------------------------------------------
// K7: compositing forward — GUTKBufferRenderer::evalKBuffer, K = 0 (gutKBufferRenderer.cuh:273-352)
// Rounds are aligned to multiples of 64 in the global sorted list, so every segment boundary (multiple of
// kGutSegment) is a round start, where the running state is checkpointed for the gradient sweep.
// ---------------------------------------------------------------------------------------------
#ifndef GRUT_FWD_PRIO
#define GRUT_FWD_PRIO 0   // K > 0: s_setprio 1 / 2 / 3 after K / 2 K / 3 K composited rounds of 64 entries (see render_fwd_sweep)
#endif
#ifndef GRUT_FWD_HALF_TILE_CULL
#define GRUT_FWD_HALF_TILE_CULL 1   // staged entries whose sphere misses the wave's ray pyramid are dropped (see WavePyramid)
#endif
struct FwdState {
    v2f T, D, Cr, Cg, Cb, cnt;
    unsigned long long t_stage;   // instrumented build: ticks spent staging rounds
};
template <int DEG, bool CKPT, bool UNI, bool COUNT = false>
__device__ __forceinline__ void render_fwd_sweep(const GutParams& P, const RayPair& rp, uint2 range, uint32_t half, int lane,
                                                 const EntryLists& lists, const float4* __restrict__ density12,
                                                 const float* __restrict__ rgb, const GutCheckpoints& ck, float4* __restrict__ s_rec,
                                                 FwdState& st) {
    bool alive0 = rp.valid0, alive1 = rp.valid1;
    v2f T = splat(1.f), D = splat(0.f), Cr = splat(0.f), Cg = splat(0.f), Cb = splat(0.f), cnt = splat(0.f);
    uint32_t n_eval = 0u, n_acc = 0u, n_rounds = 0u;   // wave-uniform work counters (scalar registers), reported when P.work is set
    const unsigned long long t_sweep = COUNT ? wall_clock64() : 0ull;
    unsigned long long t_first = 0ull, t_stage = 0ull;
    uint32_t b = range.x;
    uint32_t n_prio = 0u;   // GRUT_FWD_PRIO
    RawEntry next = load_entry<false>(b + lane, min(range.y, (b & ~63u) + 64u), lists, density12, rgb);
    constexpr bool CULL = GRUT_FWD_HALF_TILE_CULL != 0;
    WavePyramid pyr;
    if (UNI && CULL) pyr = wave_pyramid(rp);
    while (b < range.y) {
        if (!__any(alive0 || alive1)) break;
        const uint32_t bend = min(range.y, (b & ~63u) + 64u);
        if (CKPT && b > range.x && (b % kGutSegment) == 0) {
            const size_t slot = (((size_t)(b / kGutSegment) * 2 + half) * 64 + lane) * 2;
            // dead pixels restart dead (T = 0 < min_transmittance)
            ck.tc[slot] = make_float4(alive0 ? T.x : 0.f, Cr.x, Cg.x, Cb.x);
            ck.tc[slot + 1] = make_float4(alive1 ? T.y : 0.f, Cr.y, Cg.y, Cb.y);
            ck.d[slot] = D.x;
            ck.d[slot + 1] = D.y;
            if (lane == 0) ck.reached[(size_t)(b / kGutSegment) * 2 + half] = 3;   // (one bit per quarter of the half tile: both arrive together here)
        }
        int n = (int)(bend - b);
        if (COUNT) ++n_rounds;
        const unsigned long long t_round = COUNT ? wall_clock64() : 0ull;
        if (UNI && CULL) {
            // stage to a private record, drop it if its sphere misses the wave's pyramid, pack the survivors (order kept)
            float4 q[kRecQuads];
            stage_entry<DEG, false>(P, next, true, rp.origin, q);
            const bool keep = lane < n && next.idx != 0xFFFFFFFFu && !(pyr.on && pyramid_misses(pyr, q[0], q[1], q[2], q[4].w, q[5]));
            const unsigned long long km = __ballot(keep);
            if (keep) {
                float4* dst = &s_rec[__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u)) * kRecQuads];
#pragma unroll
                for (int k = 0; k < kRecQuads; ++k) dst[k] = q[k];
            }
            n = __popcll(km);
        } else {
            stage_entry<DEG, false>(P, next, UNI, rp.origin, &s_rec[lane * kRecQuads]);
        }
        __syncthreads();  // single-wave workgroup: orders the LDS hand-off
        if (COUNT && n_rounds == 1u) t_first = wall_clock64() - t_sweep;
        if (COUNT) t_stage += wall_clock64() - t_round;
        // fetch the following round while this one is being composited
        next = load_entry<false>(bend + lane, min(range.y, bend + 64u), lists, density12, rgb);
#if GRUT_FWD_MFMA
        if constexpr (UNI) {
#if GRUT_FWD_MFMA == 1
            GroupV gn = group_directions(rp, s_rec, 0, lane);
#endif
            bool stop = false;
            for (int jg = 0; jg < n && !stop; jg += 4) {
#if GRUT_FWD_MFMA == 1
                const GroupV gv = gn;
                if (jg + 4 < n) gn = group_directions(rp, s_rec, jg + 4, lane);   // in flight on the matrix pipe while this group is composited
#else
                const GroupV gv = group_directions(rp, s_rec, jg, lane);           // (2: no group ahead - 24 registers fewer; the other waves of the SIMD cover the latency)
#endif
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = jg + r;
                    if (j >= n) break;
                    if (!__any(alive0 || alive1)) { stop = true; break; }
                    const float4* rec = &s_rec[j * kRecQuads];
                    const PairGeom g = pair_geometry_given_v(p3{v2f{gv.x0[r], gv.x1[r]}, v2f{gv.y0[r], gv.y1[r]}, v2f{gv.z0[r], gv.z1[r]}}, rec);
                    const bool c0 = g.acc0 && alive0, c1 = g.acc1 && alive1;
                    if (COUNT) ++n_eval;
                    if (!__any(c0 || c1)) continue;
                    if (COUNT) ++n_acc;
                    const float4 r3 = rec[3], r4 = rec[4];
                    const v2f il2 = prcp(g.l2);
                    const v2f gray = g.cc * il2;
                    const v2f resp = pair_response<DEG>(gray);
                    const v2f ad = resp * r3.w;
                    // hit distance |S n (n.-u)| = |v.u| |S v| / |v|^2   (gaussianParticles.slang:181-190)
                    const v2f vu = pdot(g.v, g.u);
                    const p3 sv = p3{r3.x * g.v.x, r3.y * g.v.y, r3.z * g.v.z};
                    const v2f ss = pdot(sv, sv) * (vu * vu);
                    const v2f hitT = v2f{__builtin_amdgcn_sqrtf(ss.x), __builtin_amdgcn_sqrtf(ss.y)} * il2;
                    const bool h0 = c0 && (hitT.x > rp.tmin.x) && (hitT.x < rp.tmax.x);
                    const bool h1 = c1 && (hitT.y > rp.tmin.y) && (hitT.y < rp.tmax.y);
                    const v2f alpha = psel(h0, h1, v2f{fminf(P.max_alpha, ad.x), fminf(P.max_alpha, ad.y)}, splat(0.f));
                    const v2f hT = psel(h0, h1, hitT, splat(0.f));
                    const v2f w = alpha * T;
                    D = pfma(hT, w, D);
                    T = T * (1.f - alpha);
                    Cr = pfma(r4.x, w, Cr);
                    Cg = pfma(r4.y, w, Cg);
                    Cb = pfma(r4.z, w, Cb);
                    cnt += psel(w.x > 0.f, w.y > 0.f, splat(1.f), splat(0.f));
                    alive0 = alive0 && !(T.x < P.min_transmittance);
                    alive1 = alive1 && !(T.y < P.min_transmittance);
                }
            }
        } else
#endif
        for (int j = 0; j < n; ++j) {
            if (!__any(alive0 || alive1)) break;   // the rest of the round is behind every pixel's termination
            const float4* rec = &s_rec[j * kRecQuads];
            const PairGeom g = pair_geometry<UNI>(rp, rec);
            const bool c0 = g.acc0 && alive0, c1 = g.acc1 && alive1;
            if (COUNT) ++n_eval;
            if (!__any(c0 || c1)) continue;
            if (COUNT) ++n_acc;
            const float4 r3 = rec[3], r4 = rec[4];
            const v2f il2 = prcp(g.l2);
            const v2f gray = g.cc * il2;
            const v2f resp = pair_response<DEG>(gray);
            const v2f ad = resp * r3.w;
            // hit distance |S n (n.-u)| = |v.u| |S v| / |v|^2   (gaussianParticles.slang:181-190)
            const v2f vu = pdot(g.v, g.u);
            const p3 sv = p3{r3.x * g.v.x, r3.y * g.v.y, r3.z * g.v.z};
            const v2f ss = pdot(sv, sv) * (vu * vu);
            const v2f hitT = v2f{__builtin_amdgcn_sqrtf(ss.x), __builtin_amdgcn_sqrtf(ss.y)} * il2;
            const bool h0 = c0 && (hitT.x > rp.tmin.x) && (hitT.x < rp.tmax.x);
            const bool h1 = c1 && (hitT.y > rp.tmin.y) && (hitT.y < rp.tmax.y);
            const v2f alpha = psel(h0, h1, v2f{fminf(P.max_alpha, ad.x), fminf(P.max_alpha, ad.y)}, splat(0.f));
            const v2f hT = psel(h0, h1, hitT, splat(0.f));
            const v2f w = alpha * T;
            D = pfma(hT, w, D);
            T = T * (1.f - alpha);
            Cr = pfma(r4.x, w, Cr);
            Cg = pfma(r4.y, w, Cg);
            Cb = pfma(r4.z, w, Cb);
            cnt += psel(w.x > 0.f, w.y > 0.f, splat(1.f), splat(0.f));
            alive0 = alive0 && !(T.x < P.min_transmittance);
            alive1 = alive1 && !(T.y < P.min_transmittance);
        }
        __syncthreads();
        b = bend;
#if GRUT_FWD_PRIO > 0
        // a wave that has already composited many rounds is one of the launch's long ones (the lifetimes follow the accepted entries, which
        // nothing known before the launch predicts): it takes issue priority over the younger waves of its SIMD, so the long waves end
        // earlier and the launch's ramp-down shortens
        ++n_prio;
        if (n_prio == 1u * GRUT_FWD_PRIO) __builtin_amdgcn_s_setprio(1);
        else if (n_prio == 2u * GRUT_FWD_PRIO) __builtin_amdgcn_s_setprio(2);
        else if (n_prio == 3u * GRUT_FWD_PRIO) __builtin_amdgcn_s_setprio(3);
#endif
    }
    st = FwdState{T, D, Cr, Cg, Cb, cnt, 0ull};
    if (COUNT && P.work && lane == 0) {   // per-wave words, summed on the host (atomics on two shared counters would serialise the waves' exits)
        P.work[16 + 4 * (size_t)blockIdx.x + 2] = ((unsigned long long)n_acc << 32) | n_eval;
        st.t_stage = t_stage;
        // [0, 24) list length, [24, 34) staged rounds, [34, 49) first round staged, [49, 64) sweep over - both in 10 ns ticks since the sweep began
        const unsigned long long t_done = wall_clock64() - t_sweep;
        P.work[16 + 4 * (size_t)blockIdx.x + 3] = (unsigned long long)min(range.y - range.x, 0xFFFFFFu) | ((unsigned long long)min(n_rounds, 1023u) << 24) |
                                                  (min(t_first, 32767ull) << 34) | (min(t_done, 32767ull) << 49);
    }
}

#ifndef GRUT_FWD_WAVES
#define GRUT_FWD_WAVES 6   // held to 80 VGPRs (12 B of scratch): r02v 0.485 -> 0.472 ms; 0 = the allocator's own choice (90 VGPRs, 5 waves): 4 waves 0.508
#endif
template <int DEG, bool CKPT, bool COUNT = false>
__global__ __launch_bounds__(64)
#if GRUT_FWD_WAVES > 0
__attribute__((amdgpu_waves_per_eu(GRUT_FWD_WAVES, GRUT_FWD_WAVES)))
#endif
void gut_render_fwd_kernel(GutParams P, const uint2* __restrict__ ranges, EntryLists lists,
                                                            const float4* __restrict__ density12, const float* __restrict__ rgb,
                                                            const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                            float4* __restrict__ out_fd, float* __restrict__ out_dist,
                                                            float* __restrict__ out_cnt, GutCheckpoints ck) {
    __shared__ float4 s_rec[64 * kRecQuads];
    uint32_t tile, half;
    half_mapping(blockIdx.x, tile, half);
    if (tile >= (uint32_t)(P.gx * P.gy)) return;
#ifndef GRUT_FWD_TILE_STRIDE
#define GRUT_FWD_TILE_STRIDE 997   // tiles visited with a stride: r02t/r02u rows only, stride 21 of 68: 0.517 -> 0.490 ms; r02ah every tile,
#endif                             // stride 997 of 8160: another 0.017 ms (strides 61 / 499 / 1777 / 3001: no better than the rows)
    if (GRUT_FWD_TILE_STRIDE > 1) tile = stride_permute(tile, (uint32_t)(P.gx * P.gy), GRUT_FWD_TILE_STRIDE);
    const int lane = threadIdx.x;
    const unsigned long long t_begin = COUNT ? wall_clock64() : 0ull;   // constant-rate (100 MHz) counter shared by the whole chip
    const RayPair rp = init_ray_pair(P, ray_o, ray_d, tile, half, lane);
    const uint2 range = ranges[tile];
    const unsigned long long t_rays = COUNT ? wall_clock64() : 0ull;
    FwdState st;
    // two copies of the sweep: the shared-origin one keeps the canonical origin out of the per-pixel math
    if (rp.uniform_origin) render_fwd_sweep<DEG, CKPT, true, COUNT>(P, rp, range, half, lane, lists, density12, rgb, ck, s_rec, st);
    else render_fwd_sweep<DEG, CKPT, false, COUNT>(P, rp, range, half, lane, lists, density12, rgb, ck, s_rec, st);
    if (COUNT && P.work && lane == 0) {   // diagnostics: this wave's lifetime (shader-clock ticks) and start time, for the balance analysis
        const unsigned long long t_end = wall_clock64();
        // lifetime | rays ready << 32 | time spent staging << 48 (10 ns ticks)
        P.work[16 + 4 * (size_t)blockIdx.x] = (t_end - t_begin) | (min(t_rays - t_begin, 0xFFFFull) << 32) | (min(st.t_stage, 0xFFFFull) << 48);
        P.work[16 + 4 * (size_t)blockIdx.x + 1] = t_begin;
    }
    // every pixel of the image is written (the caller does not pre-fill): rays that miss the scene box get the reference's
    // initial values (splatRaster.cpp:211-214)
    if (rp.inside0) {
        const size_t pix = (size_t)rp.py0 * P.W + rp.px;
        const float4 o = rp.valid0 ? make_float4(st.Cr.x, st.Cg.x, st.Cb.x, 1.f - st.T.x) : make_float4(0.f, 0.f, 0.f, 0.f);
        store_fd(P, out_fd, pix, o);
        write_split_outputs(P, pix, o);
        out_dist[pix] = rp.valid0 ? st.D.x : 1e6f;
        if (P.hitcounts) out_cnt[pix] = rp.valid0 ? st.cnt.x : 0.f;
    }
    if (rp.inside1) {
        const size_t pix = (size_t)rp.py1 * P.W + rp.px;
        const float4 o = rp.valid1 ? make_float4(st.Cr.y, st.Cg.y, st.Cb.y, 1.f - st.T.y) : make_float4(0.f, 0.f, 0.f, 0.f);
        store_fd(P, out_fd, pix, o);
        write_split_outputs(P, pix, o);
        out_dist[pix] = rp.valid1 ? st.D.y : 1e6f;
        if (P.hitcounts) out_cnt[pix] = rp.valid1 ? st.cnt.y : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// K7q: the forward for launches that fit the chip at once (round 4).  When every wave of the launch is resident from the start
// (BASELINE configs 1 and 2: 1 250 / 5 000 half tiles for 6 144 slots) the longest wave IS the kernel, and a lone wave pays ~8 cycles per
// dependent instruction whatever its lane count: what shortens it is fewer instructions per entry.  One pixel per lane - a QUARTER tile
// (16 x 4 pixels) per wave, the .x pixels (quarter 0) or the .y pixels (quarter 1) of the half-tile wave's lanes - runs the same entry in
// ~0.6x the instructions (plain instead of packed arithmetic), walks the same list only as far as ITS 64 pixels need, and culls staged
// entries against a pyramid half as high.  Every expression below is the pair sweep's, component by component (same fused
// multiply-adds, same order): for frames with one ray origin - every frame of the static cameras - images, hit counts, checkpoints and
// with them all gradients are bit-identical to the half-tile kernel's; with per-pixel ray origins the two builds differ by an ulp per hit
// somewhere the source does not show (1.2e-7 median on the image; tests/test_gut_gpu.py::test_quarter_tile_forward_equals_the_half_tile_forward).  Checkpoints keep the pair layout
// (slot + quarter); `reached` carries one bit per quarter, set atomically, and the gradient sweep starts the pixels of a quarter that
// did not arrive dead.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float qdot(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
template <int DEG>
__device__ __forceinline__ float response1(float g) {
    constexpr float kLog2e = 1.4426950408889634f;
    if constexpr (DEG == 2) return __builtin_amdgcn_exp2f(g * (-0.5f * kLog2e));
    else if constexpr (DEG == 4) return __builtin_amdgcn_exp2f((g * g) * (-0.0555555555556f * kLog2e));
    else return particle_response<DEG>(g);
}
template <int DEG, bool CKPT, bool UNI>
__device__ __forceinline__ void render_fwd_sweep_quarter(const GutParams& P, const Ray& ry, f3 origin, uint2 range, uint32_t half, uint32_t quarter, int lane,
                                                         const EntryLists& lists, const float4* __restrict__ density12, const float* __restrict__ rgb,
                                                         const GutCheckpoints& ck, float4* __restrict__ s_rec, float (&st)[6]) {
    bool alive = ry.valid;
    float T = 1.f, D = 0.f, Cr = 0.f, Cg = 0.f, Cb = 0.f, cnt = 0.f;
    uint32_t b = range.x;
    RawEntry next = load_entry<false>(b + lane, min(range.y, (b & ~63u) + 64u), lists, density12, rgb);
    WavePyramid pyr;
    if (UNI) pyr = wave_pyramid(ry.valid, ry.d, false, ry.d);
    while (b < range.y) {
        if (!__any(alive)) break;
        const uint32_t bend = min(range.y, (b & ~63u) + 64u);
        if (CKPT && b > range.x && (b % kGutSegment) == 0) {
            const size_t idx = (size_t)(b / kGutSegment) * 2 + half;
            const size_t slot = (idx * 64 + lane) * 2 + quarter;
            ck.tc[slot] = make_float4(alive ? T : 0.f, Cr, Cg, Cb);   // dead pixels restart dead
            ck.d[slot] = D;
            if (lane == 0) atomicOr(reinterpret_cast<uint32_t*>(ck.reached) + (idx >> 2), (1u << quarter) << (8u * (uint32_t)(idx & 3)));
        }
        int n = (int)(bend - b);
        if (UNI) {
            float4 q[kRecQuads];
            stage_entry<DEG, false>(P, next, true, origin, q);
            const bool keep = lane < n && next.idx != 0xFFFFFFFFu && !(pyr.on && pyramid_misses(pyr, q[0], q[1], q[2], q[4].w, q[5]));
            const unsigned long long km = __ballot(keep);
            if (keep) {
                float4* dst = &s_rec[__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u)) * kRecQuads];
#pragma unroll
                for (int k = 0; k < kRecQuads; ++k) dst[k] = q[k];
            }
            n = __popcll(km);
        } else {
            stage_entry<DEG, false>(P, next, false, origin, &s_rec[lane * kRecQuads]);
        }
        __syncthreads();
        next = load_entry<false>(bend + lane, min(range.y, bend + 64u), lists, density12, rgb);
        for (int j = 0; j < n; ++j) {
            if (!__any(alive)) break;
            const float4* rec = &s_rec[j * kRecQuads];
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
            const f3 m0 = mk3(r0.x, r0.y, r0.z), m1 = mk3(r1.x, r1.y, r1.z), m2 = mk3(r2.x, r2.y, r2.z);
            const f3 v = mk3(qdot(m0, ry.d), qdot(m1, ry.d), qdot(m2, ry.d));
            f3 u;
            if (UNI) { const float4 r5 = rec[5]; u = mk3(r5.x, r5.y, r5.z); }
            else { const f3 dl = mk3(ry.o.x - r0.w, ry.o.y - r1.w, ry.o.z - r2.w); u = mk3(qdot(m0, dl), qdot(m1, dl), qdot(m2, dl)); }
            const float l2 = qdot(v, v);
            const f3 c = mk3(fmaf(v.y, u.z, -(v.z * u.y)), fmaf(v.z, u.x, -(v.x * u.z)), fmaf(v.x, u.y, -(v.y * u.x)));
            const float cc = qdot(c, c);
            const float4 r4 = rec[4];
            const bool acc = cc < r4.w * l2;
            const bool ca = acc && alive;
            if (!__any(ca)) continue;
            const float4 r3 = rec[3];
            const float il2 = __builtin_amdgcn_rcpf(l2);
            const float gray = cc * il2;
            const float resp = response1<DEG>(gray);
            const float ad = resp * r3.w;
            const float vu = qdot(v, u);
            const f3 sv = mk3(r3.x * v.x, r3.y * v.y, r3.z * v.z);
            const float ss = qdot(sv, sv) * (vu * vu);
            const float hitT = __builtin_amdgcn_sqrtf(ss) * il2;
            const bool h = ca && (hitT > ry.tmin) && (hitT < ry.tmax);
            const float alpha = h ? fminf(P.max_alpha, ad) : 0.f;
            const float hT = h ? hitT : 0.f;
            const float w = alpha * T;
            D = fmaf(hT, w, D);
            T = T * (1.f - alpha);
            Cr = fmaf(r4.x, w, Cr);
            Cg = fmaf(r4.y, w, Cg);
            Cb = fmaf(r4.z, w, Cb);
            cnt += w > 0.f ? 1.f : 0.f;
            alive = alive && !(T < P.min_transmittance);
        }
        __syncthreads();
        b = bend;
    }
    st[0] = T; st[1] = D; st[2] = Cr; st[3] = Cg; st[4] = Cb; st[5] = cnt;
}
// block -> (virtual tile, half, quarter): the four waves of a tile on one XCD, like half_mapping
template <int DEG, bool CKPT>
__global__ __launch_bounds__(64) void gut_render_fwd_quarter_kernel(GutParams P, const uint2* __restrict__ ranges, EntryLists lists,
                                                                    const float4* __restrict__ density12, const float* __restrict__ rgb,
                                                                    const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                                    float4* __restrict__ out_fd, float* __restrict__ out_dist,
                                                                    float* __restrict__ out_cnt, GutCheckpoints ck) {
    __shared__ float4 s_rec[64 * kRecQuads];
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    uint32_t tile = ((slot >> 2) << 3) + xcd;
    const uint32_t half = (slot >> 1) & 1u, quarter = slot & 1u;
    if (tile >= (uint32_t)(P.gx * P.gy)) return;
    if (GRUT_FWD_TILE_STRIDE > 1) tile = stride_permute(tile, (uint32_t)(P.gx * P.gy), GRUT_FWD_TILE_STRIDE);
    const int lane = threadIdx.x;
    const int px = (int)(tile % P.gx) * 16 + (lane & 15), py = (int)(tile / P.gx) * 16 + (int)half * 8 + (int)quarter * 4 + (lane >> 4);
    const Ray ry = init_ray(P, ray_o, ray_d, px, py);
    // wave-uniform origin?  (as init_ray_pair)
    const unsigned long long m = __ballot(ry.valid);
    f3 cand = mk3(0.f, 0.f, 0.f);
    if (m) {
        const int src = __ffsll((long long)m) - 1;
        cand = mk3(__shfl(ry.o.x, src, 64), __shfl(ry.o.y, src, 64), __shfl(ry.o.z, src, 64));
    }
    const bool uniform_origin = __all(!ry.valid || (ry.o.x == cand.x && ry.o.y == cand.y && ry.o.z == cand.z));
    const uint2 range = ranges[tile];
    float st[6];
    if (uniform_origin) render_fwd_sweep_quarter<DEG, CKPT, true>(P, ry, cand, range, half, quarter, lane, lists, density12, rgb, ck, s_rec, st);
    else render_fwd_sweep_quarter<DEG, CKPT, false>(P, ry, cand, range, half, quarter, lane, lists, density12, rgb, ck, s_rec, st);
    if (ry.inside) {
        const size_t pix = (size_t)py * P.W + px;
        const float4 o = ry.valid ? make_float4(st[2], st[3], st[4], 1.f - st[0]) : make_float4(0.f, 0.f, 0.f, 0.f);
        store_fd(P, out_fd, pix, o);
        write_split_outputs(P, pix, o);
        out_dist[pix] = ry.valid ? st[1] : 1e6f;
        if (P.hitcounts) out_cnt[pix] = ry.valid ? st[5] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// K8: compositing backward — evalBackwardNoKBuffer SH branch (gutKBufferRenderer.cuh:642-716) with
// processHitBwd (models/gaussianParticles.cuh:484-751).
//
// With u = gro, v = grdu (un-normalised), t = (v.u)/|v|^2, a = u - t v:  grayDist = |a|^2 and
//   d gray / d u = 2 a,      d gray / d v = -2 t a
// so with wg = dL/d gray:  uGrd = 2 wg a,  vGrd = -t uGrd,  B = uGrd / scale = dL/d(R^T (o - mu)),
//   dL/d rotT = B (x) (o - mu) + (-t B) (x) d = B (x) e,   e = (o - mu) - t d
//   dL/d position = -R B,     dL/d scale_i = -(1/s_i) sum_j rotT_ij M_ij,    M = sum over pixels of B (x) e
// which is the reference's chain (two cross-product backward passes, safe_normalize_bw, matmul_bw_vec/quat) collapsed.
// For uniform-origin waves  M = (sum B) (x) (o - mu) - sum (t B) (x) d  and only the second sum is reduced.
// When a depth gradient flows in (HAS_GDIST) the hit-distance terms (gaussianParticles.cuh:545-580) are added in
// their generic form: B (x) (o-mu) and Bv (x) d are accumulated separately and the direct scale term is reduced in a
// second pass.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kBwdBatch = 32;  // staged entries per round of the gradient sweep

struct BwdPixels {
    v2f T, D, Cr, Cg, Cb;           // running state
    v2f T_fin, D_fin, gT, gD;       // forward results and upstream gradients
    p3 C_fin, gC;
    bool alive0, alive1;
};
template <int DEG, bool HAS_GDIST, bool UNI, bool COUNT = false>
__device__ __forceinline__ void render_bwd_sweep(const GutParams& P, const RayPair& rp, uint32_t seg_begin, uint32_t seg_end, int lane,
                                                 uint32_t half, const EntryLists& lists, const float4* __restrict__ density12,
                                                 const float* __restrict__ rgb, const GutGradSlots& slots,
                                                 float4* __restrict__ s_rec, float* __restrict__ s_acc, float* __restrict__ s_acc2,
                                                 float* __restrict__ s_tr, const BwdPixels& px) {
    constexpr uint32_t kBatch = kBwdBatch;
    v2f T = px.T, D = px.D, Cr = px.Cr, Cg = px.Cg, Cb = px.Cb;
    const v2f T_fin = px.T_fin, D_fin = px.D_fin, gT = px.gT, gD = px.gD;
    const p3 C_fin = px.C_fin, gC = px.gC;
    bool alive0 = px.alive0, alive1 = px.alive1;
    uint32_t n_eval = 0u, n_acc = 0u;
    v2f iT = prcp(T);   // running 1 / T (a dead pixel restarts with T = 0: its reciprocal is never used, see inextT)
    uint32_t b = seg_begin;
    RawEntry next = load_entry(b + lane, min(seg_end, (b & ~(kBatch - 1u)) + kBatch), lists, density12, rgb);
    while (b < seg_end) {
        if (!__any(alive0 || alive1)) break;
        const uint32_t bend = min(seg_end, (b & ~(kBatch - 1u)) + kBatch);
        if (lane < (int)kBatch) stage_entry<DEG, true>(P, next, UNI, rp.origin, &s_rec[lane * kRecQuads]);
        __syncthreads();
        next = load_entry(bend + lane, min(seg_end, bend + kBatch), lists, density12, rgb);
        const int n = (int)(bend - b);
        uint32_t hit_entries = 0u;  // wave-uniform: staged entries with >= 1 hit in this wave
        for (int j = 0; j < n; ++j) {
            if (!__any(alive0 || alive1)) break;
            const float4* rec = &s_rec[j * kRecQuads];
            const PairGeom g = pair_geometry<UNI>(rp, rec);
            const bool h0 = g.acc0 && alive0, h1 = g.acc1 && alive1;
            if (COUNT) ++n_eval;
            if (!__any(h0 || h1)) continue;
            if (COUNT) ++n_acc;
            hit_entries |= (1u << j);
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4];
            const v2f il2 = prcp(g.l2);
            const v2f gray = g.cc * il2;
            const v2f resp = pair_response<DEG>(gray);
            const v2f ad = resp * r3.w;
            const v2f alpha = psel(h0, h1, v2f{fminf(P.max_alpha, ad.x), fminf(P.max_alpha, ad.y)}, splat(0.f));
            const v2f weight = alpha * T;
            const v2f oma = 1.f - alpha;
            const v2f nextT = oma * T;
            // 1 / nextT = (1 / T) (1 / (1 - alpha)): the running reciprocal iT is refreshed from T at every segment start,
            // which keeps its rounding drift below 1e-5 relative and saves two quarter-rate v_rcp_f32 per entry
            const v2f ioma = prcp(oma);
            const v2f inextT_raw = iT * ioma;
            const v2f inextT = psel(nextT.x <= P.min_transmittance, nextT.y <= P.min_transmittance, splat(0.f), inextT_raw);
            const v2f resTrm = psel(alpha.x < 0.999999f, alpha.y < 0.999999f, T_fin * ioma, T);
            v2f dalpha = -(resTrm * gT);  // d L / d alpha
            const p3 dc = p3{gC.x * weight, gC.y * weight, gC.z * weight};
            Cr = pfma(r4.x, weight, Cr); Cg = pfma(r4.y, weight, Cg); Cb = pfma(r4.z, weight, Cb);
            const v2f rr = pmax0((C_fin.x - Cr) * inextT), rg = pmax0((C_fin.y - Cg) * inextT), rb = pmax0((C_fin.z - Cb) * inextT);
            dalpha = pfma(T, pfma(r4.x - rr, gC.x, pfma(r4.y - rg, gC.y, (r4.z - rb) * gC.z)), dalpha);

            const v2f vu = pdot(g.v, g.u);
            const v2f t = vu * il2;
            const p3 a = p3{pfma(-t, g.v.x, g.u.x), pfma(-t, g.v.y, g.u.y), pfma(-t, g.v.z, g.u.z)};
            // canonical-frame offsets of the ray origin, needed un-scaled for the rotation gradient
            p3 dl;
            if (UNI) dl = p3{splat(rp.origin.x - r0.w), splat(rp.origin.y - r1.w), splat(rp.origin.z - r2.w)};
            else dl = p3{rp.o.x - r0.w, rp.o.y - r1.w, rp.o.z - r2.w};

            p3 uX = p3{splat(0.f), splat(0.f), splat(0.f)}, vX = uX, sX = uX;   // hit-distance extras
            if (HAS_GDIST) {
                // n = v/|v|, pdot = -(n.u), grds = s * n * pdot, hitT = |grds|  (gaussianParticles.cuh:545-580)
                const v2f il = v2f{__builtin_amdgcn_rsqf(g.l2.x), __builtin_amdgcn_rsqf(g.l2.y)};
                const p3 nrm = p3{g.v.x * il, g.v.y * il, g.v.z * il};
                const v2f nu = vu * il;
                const v2f pdt = -nu;
                const p3 gscl = p3{prcp(splat(r3.x)), prcp(splat(r3.y)), prcp(splat(r3.z))};
                const p3 grdd = p3{nrm.x * pdt, nrm.y * pdt, nrm.z * pdt};
                const p3 grds = p3{gscl.x * grdd.x, gscl.y * grdd.y, gscl.z * grdd.z};
                const v2f gsq = pdot(grds, grds);
                const v2f gdist = v2f{__builtin_amdgcn_sqrtf(gsq.x), __builtin_amdgcn_sqrtf(gsq.y)};
                D = pfma(weight, gdist, D);
                const v2f resHitT = pmax0((D_fin - D) * inextT);
                dalpha = pfma((gdist - resHitT) * T, gD, dalpha);
                const v2f k = psel(gsq.x > 0.f, gsq.y > 0.f, weight * prcp(gdist) * gD, splat(0.f));
                const p3 grdsGrd = p3{grds.x * k, grds.y * k, grds.z * k};
                sX = p3{grdd.x * grdsGrd.x, grdd.y * grdsGrd.y, grdd.z * grdsGrd.z};     // direct d hitT / d scale
                const p3 sg = p3{gscl.x * grdsGrd.x, gscl.y * grdsGrd.y, gscl.z * grdsGrd.z};
                const v2f sd = pdot(sg, nrm);
                const p3 nGrd = p3{pfma(sg.x, pdt, -(g.u.x * sd)), pfma(sg.y, pdt, -(g.u.y * sd)), pfma(sg.z, pdt, -(g.u.z * sd))};
                uX = p3{-(nrm.x * sd), -(nrm.y * sd), -(nrm.z * sd)};                    // d / d gro
                const v2f ng = pdot(nrm, nGrd);                                          // safe_normalize_bw
                vX = p3{(nGrd.x - nrm.x * ng) * il, (nGrd.y - nrm.y * ng) * il, (nGrd.z - nrm.z * ng) * il};
            }

            dalpha = psel(h0, h1, dalpha, splat(0.f));
            const v2f dn = resp * dalpha;
            const v2f dresp = r3.w * dalpha;
            const v2f wg2 = psel(h0, h1, 2.f * v2f{particle_response_grd<DEG>(gray.x, resp.x, dresp.x),
                                                    particle_response_grd<DEG>(gray.y, resp.y, dresp.y)}, splat(0.f));
            p3 uGrd = p3{a.x * wg2, a.y * wg2, a.z * wg2};                 // d L / d gro
            float terms[16], extra[16];
            if (!HAS_GDIST) {
                const p3 B = p3{r3.x * uGrd.x, r3.y * uGrd.y, r3.z * uGrd.z};   // d L / d (R^T (o - mu))
                p3 e;   // second factor of the rank-1 rotation gradient
                v2f s;  // ... and its scale on B
                if (UNI) { e = rp.d; s = -t; }                    // (sum B) (x) (o - mu) is added at flush
                else { e = p3{pfma(-t, rp.d.x, dl.x), pfma(-t, rp.d.y, dl.y), pfma(-t, rp.d.z, dl.z)}; s = splat(1.f); }
                const p3 Bs = p3{B.x * s, B.y * s, B.z * s};
                const v2f m[9] = {Bs.x * e.x, Bs.x * e.y, Bs.x * e.z, Bs.y * e.x, Bs.y * e.y, Bs.y * e.z, Bs.z * e.x, Bs.z * e.y, Bs.z * e.z};
                terms[0] = B.x.x + B.x.y; terms[1] = B.y.x + B.y.y; terms[2] = B.z.x + B.z.y; terms[3] = dn.x + dn.y;
#pragma unroll
                for (int k = 0; k < 9; ++k) terms[4 + k] = m[k].x + m[k].y;
                terms[13] = dc.x.x + dc.x.y; terms[14] = dc.y.x + dc.y.y; terms[15] = dc.z.x + dc.z.y;
            } else {
                const v2f mt = psel(h0, h1, splat(1.f), splat(0.f));
                uGrd = p3{pfma(uX.x, mt, uGrd.x), pfma(uX.y, mt, uGrd.y), pfma(uX.z, mt, uGrd.z)};
                const p3 vGrd = p3{pfma(vX.x, mt, -(t * a.x * wg2)), pfma(vX.y, mt, -(t * a.y * wg2)), pfma(vX.z, mt, -(t * a.z * wg2))};
                const p3 B = p3{r3.x * uGrd.x, r3.y * uGrd.y, r3.z * uGrd.z};
                const p3 Bv = p3{r3.x * vGrd.x, r3.y * vGrd.y, r3.z * vGrd.z};
                const v2f m[9] = {pfma(B.x, dl.x, Bv.x * rp.d.x), pfma(B.x, dl.y, Bv.x * rp.d.y), pfma(B.x, dl.z, Bv.x * rp.d.z),
                                  pfma(B.y, dl.x, Bv.y * rp.d.x), pfma(B.y, dl.y, Bv.y * rp.d.y), pfma(B.y, dl.z, Bv.y * rp.d.z),
                                  pfma(B.z, dl.x, Bv.z * rp.d.x), pfma(B.z, dl.y, Bv.z * rp.d.y), pfma(B.z, dl.z, Bv.z * rp.d.z)};
                terms[0] = B.x.x + B.x.y; terms[1] = B.y.x + B.y.y; terms[2] = B.z.x + B.z.y; terms[3] = dn.x + dn.y;
#pragma unroll
                for (int k = 0; k < 9; ++k) terms[4 + k] = m[k].x + m[k].y;
                terms[13] = dc.x.x + dc.x.y; terms[14] = dc.y.x + dc.y.y; terms[15] = dc.z.x + dc.z.y;
                const p3 sXm = p3{sX.x * mt, sX.y * mt, sX.z * mt};
#pragma unroll
                for (int k = 0; k < 16; ++k) extra[k] = 0.f;
                extra[0] = sXm.x.x + sXm.x.y; extra[1] = sXm.y.x + sXm.y.y; extra[2] = sXm.z.x + sXm.z.y;
            }
#if GRUT_BWD_FULL_REDUCE == 2
            // transpose through LDS instead of the DPP butterfly: the sweep is bound by VALU issue and the LDS pipe is idle, so the
            // 16 x 64 -> 16 reduction is moved there: every lane parks its 16 terms (term-major, rows padded to 65 words: conflict-free
            // both ways), lane l then sums 16 of the 64 values of term l & 15 and two lane-swap steps add the four quarters
            {
#pragma unroll
                for (int k = 0; k < 16; ++k) s_tr[k * 65 + lane] = terms[k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                const float* col = &s_tr[(lane & 15) * 65 + (lane >> 4) * 16];
                float part = col[0];
#pragma unroll
                for (int k = 1; k < 16; ++k) part += col[k];
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const v2u sa = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
                const float s2 = __uint_as_float(sa.x) + __uint_as_float(sa.y);
                const v2u sb = __builtin_amdgcn_permlane16_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
                const float tot = __uint_as_float(sb.x) + __uint_as_float(sb.y);
                if (lane < 16) s_acc[j * 16 + lane] = tot;
                __builtin_amdgcn_wave_barrier();   // the next entry overwrites s_tr
            }
            if (HAS_GDIST) {
                const float tot2 = wave_reduce_scatter16_all(extra, lane);
                if (lane < 16) s_acc2[j * 16 + lane] = tot2;
            }
#elif GRUT_BWD_FULL_REDUCE
            // every lane ends with the wave total of term l & 15 (in-row DPP butterfly, then two lane-swap steps across the rows);
            // one row of 16 lanes parks the totals of the entry in LDS for the flush
            const float tot = wave_reduce_scatter16_all(terms, lane);
            if (lane < 16) s_acc[j * 16 + lane] = tot;
            if (HAS_GDIST) {
                const float tot2 = wave_reduce_scatter16_all(extra, lane);
                if (lane < 16) s_acc2[j * 16 + lane] = tot2;
            }
#else
            // lane l ends with the sum over its 16-lane row of term l & 15; the four rows are added by the flush
            s_acc[j * 64 + lane] = wave_reduce_scatter16_rows(terms, lane);
            if (HAS_GDIST) s_acc2[j * 64 + lane] = wave_reduce_scatter16_rows(extra, lane);
#endif
            T = nextT;
            iT = inextT_raw;
            alive0 = alive0 && !(T.x < P.min_transmittance);
            alive1 = alive1 && !(T.y < P.min_transmittance);
        }
        __syncthreads();
        // flush: lane j owns staged entry j and stores the wave totals of its 16 (+3) terms to the entry's gradient slot
        if (lane < (int)kBatch && ((hit_entries >> lane) & 1u)) {
            const float4* rec = &s_rec[lane * kRecQuads];
            const uint32_t pos = __float_as_uint(rec[5].w);
#if GRUT_BWD_FULL_REDUCE
            const float4* acc = reinterpret_cast<const float4*>(&s_acc[lane * 16]);
            float4 a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
#else
            const float4* acc = reinterpret_cast<const float4*>(&s_acc[lane * 64]);
            float4 a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
#pragma unroll
            for (int r = 1; r < 4; ++r) {
                const float4 b0 = acc[4 * r], b1 = acc[4 * r + 1], b2 = acc[4 * r + 2], b3 = acc[4 * r + 3];
                a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w; a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
                a2.x += b2.x; a2.y += b2.y; a2.z += b2.z; a2.w += b2.w; a3.x += b3.x; a3.y += b3.y; a3.z += b3.z; a3.w += b3.w;
            }
#endif
            if (!HAS_GDIST && UNI) {  // complete M = (sum B) (x) (o - mu) - sum (t B) (x) d
                const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                const f3 dl = rp.origin - mk3(r0.w, r1.w, r2.w);
                a1.x += a0.x * dl.x; a1.y += a0.x * dl.y; a1.z += a0.x * dl.z;
                a1.w += a0.y * dl.x; a2.x += a0.y * dl.y; a2.y += a0.y * dl.z;
                a2.z += a0.z * dl.x; a2.w += a0.z * dl.y; a3.x += a0.z * dl.z;
            }
            const size_t slot = 2 * (size_t)pos + half;
            float4* out = reinterpret_cast<float4*>(slots.partial + slot * (HAS_GDIST ? 20 : 16));
            out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3;
            if (HAS_GDIST) {
                float ex = 0.f, ey = 0.f, ez = 0.f;
#if GRUT_BWD_FULL_REDUCE
                ex = s_acc2[lane * 16]; ey = s_acc2[lane * 16 + 1]; ez = s_acc2[lane * 16 + 2];
#else
                for (int r = 0; r < 4; ++r) { ex += s_acc2[lane * 64 + 16 * r]; ey += s_acc2[lane * 64 + 16 * r + 1]; ez += s_acc2[lane * 64 + 16 * r + 2]; }
#endif
                out[4] = make_float4(ex, ey, ez, 0.f);
            }
            slots.flag[slot] = 1;
        }
        __syncthreads();
        b = bend;
    }
    if (COUNT && P.work && lane == 0) { atomicAdd(&P.work[2], (unsigned long long)n_eval); atomicAdd(&P.work[3], (unsigned long long)n_acc); }
}

template <int DEG, bool HAS_GDIST, bool COUNT = false>
#if GRUT_BWD_WAVES   // the depth-gradient variant (not the training path) needs ~160 registers: it keeps its 3 waves per SIMD
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HAS_GDIST ? 3 : GRUT_BWD_WAVES, HAS_GDIST ? 3 : GRUT_BWD_WAVES))) void gut_render_bwd_kernel(
#else
__global__ __launch_bounds__(64) void gut_render_bwd_kernel(
#endif
                                                            GutParams P, const uint2* __restrict__ ranges, EntryLists lists,
                                                            const float4* __restrict__ density12, const float* __restrict__ rgb,
                                                            const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                            const float4* __restrict__ fd, GutGradIn g_in,
                                                            const float* __restrict__ dist, const float* __restrict__ g_dist,
                                                            GutGradSlots slots, GutCheckpoints ck) {
    constexpr uint32_t kBatch = kBwdBatch;
    __shared__ float4 s_rec[kBatch * kRecQuads];
#if GRUT_BWD_FULL_REDUCE == 2
    __shared__ float s_tr[16 * 65];                            // transposition buffer of the per-entry reduction
#else
    float* s_tr = nullptr;
#endif
#if GRUT_BWD_FULL_REDUCE
    __shared__ float s_acc[kBatch * 16];                       // per staged entry: the wave totals of its 16 terms
    __shared__ float s_acc2[HAS_GDIST ? kBatch * 16 : 1];      // depth-gradient extras (3 terms used)
#else
    __shared__ float s_acc[kBatch * 64];                       // per staged entry: 16 terms x 4 row partials
    __shared__ float s_acc2[HAS_GDIST ? kBatch * 64 : 1];      // depth-gradient extras (3 terms used)
#endif
    // task = (virtual tile, half).  Virtual tiles [0, bndPad) are the segments that start at segment boundary b (sorted
    // index b * kGutSegment): they are the long, dense tasks and are dispatched first so that the tail of the launch
    // is made of the short first segments of each tile list, virtual tiles [bndPad, bndPad + tiles).  Only a quarter of the
    // launched workgroups find work on the bench frame (the forward reached 26 k of the ~106 k possible tasks alive); compacting
    // them into a task list first (persistent workers, or one workgroup per real task) was measured and is NOT faster: the empty
    // workgroups retire in the shadow of the running ones (DESIGN.md "Gradient sweep: task dispatch").  Measured again in round 4 with 64-entry
    // segments (376 k workgroups for ~100 k tasks; launching them all EMPTY takes 82 us): a compacted task list built by a small kernel, the grid
    // sized from the previous frame's count - gradient sweep 0.825-0.829 ms against 0.810-0.816 ms with the direct mapping.  Not kept.
    const uint32_t num_tiles = (uint32_t)(P.gx * P.gy), bnd_pad = (ck.num_boundaries + 7u) & ~7u;
    uint32_t vtile, half;
    half_mapping(blockIdx.x, vtile, half);
    uint32_t tile, seg_begin;
    bool from_checkpoint = false;
    uint32_t boundary = 0, reached_bits = 3u;   // bit q: quarter q of the half tile (pixel .x / .y of the lanes) arrived at the boundary alive
#ifndef GRUT_BWD_TASK_STRIDE
#define GRUT_BWD_TASK_STRIDE 0   // (a strided order of the gradient sweep's tasks was measured slower: 0.854-0.863 vs 0.839 ms)
#endif
    if (vtile >= bnd_pad) {
        tile = vtile - bnd_pad;
        if (tile >= num_tiles) return;
        seg_begin = ranges[tile].x;
    } else {
        boundary = GRUT_BWD_TASK_STRIDE > 1 ? stride_permute(vtile, bnd_pad, GRUT_BWD_TASK_STRIDE) : vtile;
        if (boundary == 0 || boundary >= ck.num_boundaries) return;
        reached_bits = ck.reached[(size_t)boundary * 2 + half];
        if (!reached_bits) return;                                   // the forward sweep never got here alive
        tile = ck.boundary_tile[boundary];
        if (tile >= num_tiles) return;
        seg_begin = boundary * kGutSegment;
        if (seg_begin <= ranges[tile].x) return;                     // the boundary is this tile's own list start
        from_checkpoint = true;
    }
    const int lane = threadIdx.x;
    const uint32_t seg_end = min(ranges[tile].y, (seg_begin / kGutSegment + 1u) * kGutSegment);
    const unsigned long long t_begin = COUNT ? wall_clock64() : 0ull;
    const RayPair rp = init_ray_pair(P, ray_o, ray_d, tile, half, lane);
    bool alive0 = rp.valid0, alive1 = rp.valid1;

    v2f T = splat(1.f), D = splat(0.f), Cr = splat(0.f), Cg = splat(0.f), Cb = splat(0.f);
    v2f T_fin = splat(0.f), D_fin = splat(0.f), gT = splat(0.f), gD = splat(0.f);
    p3 C_fin = p3{splat(0.f), splat(0.f), splat(0.f)}, gC = C_fin;
    if (alive0) {
        const size_t pix = (size_t)rp.py0 * P.W + rp.px;
        const float4 f = load_fd(P, fd, pix), g = load_grad_in(g_in, pix);
        C_fin.x.x = f.x; C_fin.y.x = f.y; C_fin.z.x = f.z; gC.x.x = g.x; gC.y.x = g.y; gC.z.x = g.z;
        T_fin.x = 1.f - f.w; gT.x = -g.w;
        if (HAS_GDIST) { D_fin.x = dist[pix]; gD.x = g_dist[pix]; }
    }
    if (alive1) {
        const size_t pix = (size_t)rp.py1 * P.W + rp.px;
        const float4 f = load_fd(P, fd, pix), g = load_grad_in(g_in, pix);
        C_fin.x.y = f.x; C_fin.y.y = f.y; C_fin.z.y = f.z; gC.x.y = g.x; gC.y.y = g.y; gC.z.y = g.z;
        T_fin.y = 1.f - f.w; gT.y = -g.w;
        if (HAS_GDIST) { D_fin.y = dist[pix]; gD.y = g_dist[pix]; }
    }
    if (from_checkpoint) {
        const size_t slot = (((size_t)boundary * 2 + half) * 64 + lane) * 2;
        // (a quarter-tile forward wave that was done before this boundary left no checkpoint: its pixels restart dead)
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 c0 = (reached_bits & 1u) ? ck.tc[slot] : zero4, c1 = (reached_bits & 2u) ? ck.tc[slot + 1] : zero4;
        T = v2f{c0.x, c1.x}; Cr = v2f{c0.y, c1.y}; Cg = v2f{c0.z, c1.z}; Cb = v2f{c0.w, c1.w};
        if (HAS_GDIST) D = v2f{(reached_bits & 1u) ? ck.d[slot] : 0.f, (reached_bits & 2u) ? ck.d[slot + 1] : 0.f};
        alive0 = alive0 && !(T.x < P.min_transmittance);
        alive1 = alive1 && !(T.y < P.min_transmittance);
    }

    BwdPixels px{T, D, Cr, Cg, Cb, T_fin, D_fin, gT, gD, C_fin, gC, alive0, alive1};
    if (rp.uniform_origin)
        render_bwd_sweep<DEG, HAS_GDIST, true, COUNT>(P, rp, seg_begin, seg_end, lane, half, lists, density12, rgb, slots, s_rec, s_acc, s_acc2, s_tr, px);
    else
        render_bwd_sweep<DEG, HAS_GDIST, false, COUNT>(P, rp, seg_begin, seg_end, lane, half, lists, density12, rgb, slots, s_rec, s_acc, s_acc2, s_tr, px);
    if (COUNT && P.work && lane == 0) {   // diagnostics (after the forward sweep's block of per-wave words): lifetime and start of this task
        const size_t gid = (size_t)atomicAdd(&P.work[8], 1ull);
        const size_t base = 16 + 4 * (size_t)(((num_tiles + 7u) & ~7u) * 2u) + 2 * gid;
        if (gid < P.work_task_capacity) {   // (the block was sized from an earlier frame's list length: records beyond it are dropped)
            P.work[base] = wall_clock64() - t_begin;
            P.work[base + 1] = t_begin;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K > 0 ("sorted") compositing — HitParticleKBufferT<K> + evalKBuffer (gutKBufferRenderer.cuh:62-122, 273-352) and its
// backward processHitParticle<Backward> (:158-198; Slang reverse mode of the back-to-front lerp form,
// shRadiativeParticles.slang:210-256, gaussianParticles.slang:420-479).
// Every pixel keeps the K nearest pending hits (by hitT) in registers and composites the nearest one when the buffer is
// full, so neighbouring pixels pop DIFFERENT particles at the same step: there is nothing to reduce across the wave, and
// the gradient is accumulated per hit with atomics exactly like the reference does in this mode.  One pixel per lane,
// one wave64 per 16x4 strip; correctness-first kernel (the unsorted K = 0 path above is the tuned one).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gray_limit_rt(int deg, float x) {
    switch (deg) {
    case 8: return response_gray_limit<8>(x);
    case 5: return response_gray_limit<5>(x);
    case 4: return response_gray_limit<4>(x);
    case 3: return response_gray_limit<3>(x);
    case 1: return response_gray_limit<1>(x);
    case 0: return response_gray_limit<0>(x);
    default: return response_gray_limit<2>(x);
    }
}
__device__ __forceinline__ float response_rt(int deg, float g) {
    switch (deg) {
    case 8: return particle_response<8>(g);
    case 5: return particle_response<5>(g);
    case 4: return particle_response<4>(g);
    case 3: return particle_response<3>(g);
    case 1: return particle_response<1>(g);
    case 0: return particle_response<0>(g);
    default: return particle_response<2>(g);
    }
}
__device__ __forceinline__ float response_grd_rt(int deg, float g, float gres, float gresGrd) {
    switch (deg) {
    case 8: return particle_response_grd<8>(g, gres, gresGrd);
    case 5: return particle_response_grd<5>(g, gres, gresGrd);
    case 4: return particle_response_grd<4>(g, gres, gresGrd);
    case 3: return particle_response_grd<3>(g, gres, gresGrd);
    case 1: return particle_response_grd<1>(g, gres, gresGrd);
    case 0: return particle_response_grd<0>(g, gres, gresGrd);
    default: return particle_response_grd<2>(g, gres, gresGrd);
    }
}

// Sorted mode: the k-buffer ORDERS a ray's hits by their fp32 hit distance, so two implementations agree on the order only if they agree on
// the distance's bits.  The hit distance of an accepted hit is therefore evaluated in the CHECKER's operation order (the checker's gut_oracle.c:
// density_hit_ex, itself the source order of gaussianParticles.slang:96-110, 181-190): every product and sum rounded on its own, correctly
// rounded 1/x and sqrt - ~70 instructions per ACCEPTED hit on top of the accept test, which keeps the fast pre-transformed form (its flips
// are identified per pixel by the parity tests; an order tie could not be).  rt* = rows of R^T from the quaternion, gis = 1 / scale.
__device__ __forceinline__ float sub_rn(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void rn_rotT(float r, float x, float y, float z, f3& r0, f3& r1, f3& r2) {   // quat_wxyz_to_rotT, uncontracted
    const float xx = mul_rn(x, x), yy = mul_rn(y, y), zz = mul_rn(z, z), xy = mul_rn(x, y), xz = mul_rn(x, z), yz = mul_rn(y, z);
    const float rx = mul_rn(r, x), ry = mul_rn(r, y), rz = mul_rn(r, z);
    r0 = mk3(sub_rn(1.f, mul_rn(2.f, add_rn(yy, zz))), mul_rn(2.f, add_rn(xy, rz)), mul_rn(2.f, sub_rn(xz, ry)));
    r1 = mk3(mul_rn(2.f, sub_rn(xy, rz)), sub_rn(1.f, mul_rn(2.f, add_rn(xx, zz))), mul_rn(2.f, add_rn(yz, rx)));
    r2 = mk3(mul_rn(2.f, add_rn(xz, ry)), mul_rn(2.f, sub_rn(yz, rx)), sub_rn(1.f, mul_rn(2.f, add_rn(xx, yy))));
}
__device__ __forceinline__ float oracle_order_hit_t(const Ray& ray, f3 pos, f3 scl, f3 rt0, f3 rt1, f3 rt2, f3 gis) {
    const f3 gposc = mk3(sub_rn(ray.o.x, pos.x), sub_rn(ray.o.y, pos.y), sub_rn(ray.o.z, pos.z));
    const f3 gro = mk3(mul_rn(gis.x, rn_dot3(rt0, gposc)), mul_rn(gis.y, rn_dot3(rt1, gposc)), mul_rn(gis.z, rn_dot3(rt2, gposc)));
    const f3 grdu = mk3(mul_rn(gis.x, rn_dot3(rt0, ray.d)), mul_rn(gis.y, rn_dot3(rt1, ray.d)), mul_rn(gis.z, rn_dot3(rt2, ray.d)));
    const float inv = 1.f / sqrtf(rn_dot3(grdu, grdu));            // (IEEE division and square root: the file is not built with fast-math)
    const f3 grd = mk3(mul_rn(grdu.x, inv), mul_rn(grdu.y, inv), mul_rn(grdu.z, inv));
    const float along = rn_dot3(grd, mk3(-gro.x, -gro.y, -gro.z));
    const f3 grds = mk3(mul_rn(scl.x, mul_rn(grd.x, along)), mul_rn(scl.y, mul_rn(grd.y, along)), mul_rn(scl.z, mul_rn(grd.z, along)));
    return sqrtf(rn_dot3(grds, grds));
}

template <int K>
struct KBuffer {   // ascending in hitT: slot 0 = nearest pending hit, empty slots hold hitT = -1 at the front
    float hitT[K], alpha[K];
    uint32_t idx[K];
    int num;
    __device__ __forceinline__ void clear() {
        num = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) { hitT[i] = -1.f; alpha[i] = 0.f; idx[i] = 0xFFFFFFFFu; }
    }
    // HitParticleKBufferT::insert (:76-91); the caller has already consumed slot 0 when the buffer was full
    // branch-free: the keys move with min / max (no swap on equal keys, like the reference's strict `>`), the payloads
    // with selects
    __device__ __forceinline__ void insert(float t, float a, uint32_t id) {
#pragma unroll
        for (int i = K - 1; i >= 0; --i) {
            const bool up = t > hitT[i];
            const float lo = fminf(t, hitT[i]), hi = fmaxf(t, hitT[i]);
            const float aa = up ? alpha[i] : a;
            const uint32_t ii = up ? idx[i] : id;
            alpha[i] = up ? a : alpha[i];
            idx[i] = up ? id : idx[i];
            hitT[i] = hi;
            t = lo; a = aa; id = ii;
        }
    }
};

struct KFwdState {
    float T, D, Cr, Cg, Cb, cnt;
};
struct KBwdState {
    float T;                 // forward running transmittance (termination only)
    float Tb, Db, gT, gD;    // "behind" values and running upstream gradients
    f3 Cb, gC;
};

__device__ __forceinline__ void k_process_fwd(const GutParams& P, const float* __restrict__ rgb, float hitT, float alpha, uint32_t idx,
                                              KFwdState& s, bool& alive) {
    const float w = alpha * s.T;
    s.D = fmaf(hitT, w, s.D);
    s.T *= (1.f - alpha);
    if (w > 0.f) {
        s.Cr = fmaf(fmaxf(rgb[3 * (size_t)idx], 0.f), w, s.Cr);
        s.Cg = fmaf(fmaxf(rgb[3 * (size_t)idx + 1], 0.f), w, s.Cg);
        s.Cb = fmaf(fmaxf(rgb[3 * (size_t)idx + 2], 0.f), w, s.Cb);
        s.cnt += 1.f;
    }
    if (s.T < P.min_transmittance) alive = false;
}

// gradient of sum_ij (b_i e_j) rotT_ij(q) w.r.t. q = (r,x,y,z)  (matmul_bw_quat, mathUtils.cuh:458-521)
__device__ __forceinline__ float4 quat_outer_contract(f3 b, f3 e, float4 q) {
    const float r = 2.f * q.x, x = 2.f * q.y, y = 2.f * q.z, z = 2.f * q.w;
    const float m00 = b.x * e.x, m01 = b.x * e.y, m02 = b.x * e.z, m10 = b.y * e.x, m11 = b.y * e.y, m12 = b.y * e.z, m20 = b.z * e.x,
                m21 = b.z * e.y, m22 = b.z * e.z;
    const float s01 = m01 + m10, s02 = m02 + m20, s12 = m12 + m21, a01 = m01 - m10, a02 = m20 - m02, a12 = m12 - m21;
    return make_float4(z * a01 + y * a02 + x * a12, y * s01 + z * s02 + r * a12 - 2.f * x * (m11 + m22),
                       x * s01 + z * s12 + r * a02 - 2.f * y * (m00 + m22), x * s02 + y * s12 + r * a01 - 2.f * z * (m00 + m11));
}

// One hit of one pixel in the sorted mode's backward: updates the running state and returns the hit's 14 gradient terms
// (position 0-2, density 3, quaternion 4-7, scale 8-10, rgb 11-13) instead of adding them to memory; `k_bwd_flush` adds
// them wave by wave.  Returns whether the hit contributed.
__device__ __forceinline__ bool k_bwd_terms(const GutParams& P, const Ray& ray, const float4* __restrict__ density12, const float* __restrict__ rgb,
                                            float hitT, float alpha, uint32_t idx, KBwdState& s, bool& alive, float (&terms)[16]) {
    const bool contributes = alpha > 0.f;
    if (contributes) {
        const float w = __builtin_amdgcn_rcpf(1.f - alpha);
        const f3 feat = mk3(fmaxf(rgb[3 * (size_t)idx], 0.f), fmaxf(rgb[3 * (size_t)idx + 1], 0.f), fmaxf(rgb[3 * (size_t)idx + 2], 0.f));
        s.Cb = (s.Cb - feat * alpha) * w;
        float dalpha = (feat.x - s.Cb.x) * s.gC.x + (feat.y - s.Cb.y) * s.gC.y + (feat.z - s.Cb.z) * s.gC.z;
        terms[11] = alpha * s.gC.x; terms[12] = alpha * s.gC.y; terms[13] = alpha * s.gC.z;
        s.gC = s.gC * (1.f - alpha);
        s.Tb *= w;
        s.Db = (s.Db - hitT * alpha) * w;
        dalpha += (hitT - s.Db) * s.gD - s.Tb * s.gT;
        const float ddepth = alpha * s.gD;
        s.gD *= (1.f - alpha);
        s.gT *= (1.f - alpha);

        const float4 a = density12[3 * (size_t)idx], q = density12[3 * (size_t)idx + 1], sc = density12[3 * (size_t)idx + 2];
        const m3 rotT = quat_wxyz_to_rotT(q.x, q.y, q.z, q.w);
        const f3 gscl = mk3(sc.x, sc.y, sc.z), giscl = mk3(__builtin_amdgcn_rcpf(sc.x), __builtin_amdgcn_rcpf(sc.y), __builtin_amdgcn_rcpf(sc.z));
        const f3 gposc = ray.o - mk3(a.x, a.y, a.z);
        const f3 gposcr = mul_rows(rotT, gposc);
        const f3 gro = giscl * gposcr;
        const f3 rdr = mul_rows(rotT, ray.d);
        const f3 grdu = giscl * rdr;
        const float l2 = dot(grdu, grdu);
        const float il = __builtin_amdgcn_rsqf(l2);
        const f3 grd = grdu * il;
        const f3 gcrod = cross(grd, gro);
        const float gray = dot(gcrod, gcrod);
        const float gres = response_rt(P.degree, gray);
        float dres = 0.f, ddens = 0.f;  // reverse mode of min(MaxAlpha, .): only the smaller argument receives the gradient
        if (gres * a.w < P.max_alpha) { dres = a.w * dalpha; ddens = gres * dalpha; }
        const float grayGrd = response_grd_rt(P.degree, gray, gres, dres);
        const float pdot = -dot(grd, gro);
        const f3 grdd = grd * pdot;
        const f3 grds = gscl * grdd;
        const float gsq = dot(grds, grds);
        const float gdist = __builtin_amdgcn_sqrtf(gsq);
        const f3 grdsGrd = gsq > 0.f ? grds * (ddepth * __builtin_amdgcn_rcpf(gdist)) : mk3(0.f, 0.f, 0.f);
        const f3 gsclHit = grdd * grdsGrd;
        const float sdot = dot(grdsGrd * gscl, grd);
        const f3 grdHit = gscl * grdsGrd * pdot - gro * sdot;
        const f3 groHit = grd * (-sdot);
        const f3 gcrodGrd = gcrod * (2.f * grayGrd);
        const f3 grdGrd = mk3(gcrodGrd.z * gro.y - gcrodGrd.y * gro.z, gcrodGrd.x * gro.z - gcrodGrd.z * gro.x, gcrodGrd.y * gro.x - gcrodGrd.x * gro.y);
        const f3 groGrd = mk3(gcrodGrd.y * grd.z - gcrodGrd.z * grd.y, gcrodGrd.z * grd.x - gcrodGrd.x * grd.z, gcrodGrd.x * grd.y - gcrodGrd.y * grd.x);
        const f3 groTot = groGrd + groHit;
        const f3 is2 = giscl * giscl;
        const f3 gsclGro = mk3(-gposcr.x * is2.x, -gposcr.y * is2.y, -gposcr.z * is2.z) * groTot;
        const f3 gposcrGrd = giscl * groTot;
        const f3 gposcGrd = mul_cols(rotT, gposcrGrd);
        const f3 dn = grdGrd + grdHit;
        const f3 grduGrd = dn * il - grdu * (il * il * il * dot(dn, grdu));   // normalize backward
        const f3 sclGrd = gsclHit + gsclGro + mk3(-rdr.x * is2.x, -rdr.y * is2.y, -rdr.z * is2.z) * grduGrd;
        const float4 gq1 = quat_outer_contract(gposcrGrd, gposc, q), gq2 = quat_outer_contract(giscl * grduGrd, ray.d, q);
        terms[0] = -gposcGrd.x; terms[1] = -gposcGrd.y; terms[2] = -gposcGrd.z; terms[3] = ddens;
        terms[4] = gq1.x + gq2.x; terms[5] = gq1.y + gq2.y; terms[6] = gq1.z + gq2.z; terms[7] = gq1.w + gq2.w;
        terms[8] = sclGrd.x; terms[9] = sclGrd.y; terms[10] = sclGrd.z;
    }
    s.T *= (1.f - alpha);
    if (s.T < P.min_transmittance) alive = false;
    return contributes;
}
// Adds the terms of the lanes with `have` to the gradient buffers: lanes that processed the SAME particle in this step
// (neighbouring pixels usually do) are summed with a DPP reduce-scatter first, one set of 14 atomics per (wave, particle)
// instead of one per (pixel, particle) — the reference's per-hit atomics (gutKBufferRenderer.cuh:158-198) cost this path
// 147 ms per 1080p frame on MI355X.  (A further level — a per-wave LDS cache of per-particle totals across steps, evicted
// to memory — was measured and dropped: 8.0 -> 9.6 ms; the atomics are not what bounds this kernel.)
__device__ __forceinline__ void k_bwd_flush(bool have, uint32_t idx, const float (&terms)[16], int lane, float* __restrict__ g_density12,
                                            float* __restrict__ g_rgb) {
    unsigned long long m = __ballot(have);
    while (m) {
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t pid = (uint32_t)__builtin_amdgcn_readlane((int)idx, leader);
        const bool part = have && (idx == pid);
        float t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = (part && k < 14) ? terms[k] : 0.f;
        const float tot = wave_reduce_scatter16(t, lane);
        if (lane < 11) atomicAdd(g_density12 + 12 * (size_t)pid + lane, tot);
        else if (lane < 14) atomicAdd(g_rgb + 3 * (size_t)pid + (lane - 11), tot);
        m &= ~__ballot(part);
    }
}

// Round 5: the same sums through LDS instead of the DPP reduce-scatter (GRUT_K_FLUSH_LDS, default on).  A step's popping lanes hold
// DIFFERENT particles more often than not (a pop is an OLD pending hit; neighbouring pixels' buffers drift apart), so the loop above runs
// once per distinct particle at ~100 instructions each (16 selects + the 30-step reduce-scatter + the atomics).  Here every popping lane
// parks its 14 terms in LDS (stride 15: conflict-free both ways) and lanes 0..13 ARE the 14 words of a particle's gradient: per distinct
// particle the members' words are added from LDS (one read + one add per member) and leave as ONE atomic instruction - the hit-major
// transposition of the 3DGRT replay backward (grt_replay_bwd_kernel).
#ifndef GRUT_K_FLUSH_LDS
#define GRUT_K_FLUSH_LDS 1
#endif
#ifndef GRUT_K_FLUSH_MINK
#define GRUT_K_FLUSH_MINK 8    // smallest K that takes the LDS flush and the three-waves allocation that goes with it (K = 4 runs three waves
                               // on the DPP reduce-scatter already; measured at 1 M / 1080p, step: K = 16 11.89 -> 10.90 ms, K = 8 8.89 -> 8.41 ms)
#endif
constexpr int kKTermStride = 15;
__device__ __forceinline__ void k_bwd_flush_lds(bool have, uint32_t idx, const float (&terms)[16], int lane, float* __restrict__ s_terms,
                                                float* __restrict__ g_density12, float* __restrict__ g_rgb) {
    if (have) {
#pragma unroll
        for (int k = 0; k < 14; ++k) s_terms[lane * kKTermStride + k] = terms[k];
    }
    __syncthreads();   // single-wave workgroup: orders the LDS hand-off
    unsigned long long m = __ballot(have);
    float* const row = lane < 11 ? g_density12 + lane : g_rgb + (lane < 14 ? lane - 11 : 0);
    const uint32_t stride = lane < 11 ? 12u : 3u;
    while (m) {
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t pid = (uint32_t)__builtin_amdgcn_readlane((int)idx, leader);
        unsigned long long same = __ballot(have && idx == pid);
        m &= ~same;
        float v = 0.f;
        while (same) {
            const int s2 = __ffsll((long long)same) - 1;
            same &= same - 1;
            v += s_terms[s2 * kKTermStride + (lane < 14 ? lane : 0)];
        }
        if (lane < 14 && v != 0.f) atomicAdd(row + (size_t)pid * stride, v);
    }
    __syncthreads();   // the next step overwrites s_terms.  (Two buffers used alternately - one barrier per step - were measured SLOWER: 15.9 KB of
                       // LDS per wave leave room for 10 waves per CU, i.e. two per SIMD again: backward 9.5 instead of 7.6 ms.)
}

template <int K, bool BWD>
__device__ __forceinline__ void gut_render_k_body(const GutParams& P, const uint2* __restrict__ ranges, const EntryLists& lists,
                                                  const float4* __restrict__ density12, const float* __restrict__ rgb,
                                                  const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                  float4* __restrict__ out_fd, float* __restrict__ out_dist, float* __restrict__ out_cnt,
                                                  const float4* __restrict__ g_fd, const float* __restrict__ g_dist,
                                                  float* __restrict__ g_density12, float* __restrict__ g_rgb) {
    constexpr int kQ = 8;   // quads per staged entry: 0-2 M rows | pos, 3 scale | density, 4 particle | accept limit, 5-7 rows of R^T | 1 / scale
    __shared__ float4 s_rec[64 * kQ];
    constexpr bool kLdsFlush = BWD && GRUT_K_FLUSH_LDS && K >= GRUT_K_FLUSH_MINK;
    __shared__ float s_kterms[kLdsFlush ? 64 * kKTermStride : 1];
    // strip -> (tile, strip-in-tile) with all four strips of a tile on one XCD
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t tile = ((slot >> 2) << 3) + xcd, strip = slot & 3u;
    if (tile >= (uint32_t)(P.gx * P.gy)) return;
    const int lane = threadIdx.x;
    const int px = (int)(tile % P.gx) * 16 + (lane & 15);
    const int py = (int)(tile / P.gx) * 16 + (int)strip * 4 + (lane >> 4);
    const Ray ray = init_ray(P, ray_o, ray_d, px, py);
    bool alive = ray.valid;
    const size_t pix = ray.valid ? (size_t)py * P.W + px : 0;
    KFwdState fs{1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    KBwdState bs;
    bs.T = 1.f; bs.Tb = 0.f; bs.Db = 0.f; bs.gT = 0.f; bs.gD = 0.f; bs.Cb = mk3(0.f, 0.f, 0.f); bs.gC = mk3(0.f, 0.f, 0.f);
    if (BWD && alive) {   // initializeBackwardRay (rayPayloadBackward.cuh:30-73); out_* hold the forward results here
        const float4 f = load_fd(P, out_fd, pix), g = g_fd[pix];
        bs.Cb = mk3(f.x, f.y, f.z); bs.gC = mk3(g.x, g.y, g.z);
        bs.Tb = 1.f - f.w; bs.gT = -g.w;
        bs.Db = out_dist[pix]; bs.gD = g_dist ? g_dist[pix] : 0.f;
    }
    KBuffer<K> kb;
    kb.clear();
    const uint2 range = ranges[tile];
    for (uint32_t b = range.x; b < range.y; b += 64) {
        if (!__any(alive)) break;
        {   // stage up to 64 entries
            const RawEntry e = load_entry(b + lane, range.y, lists, density12, rgb);
            float4 r0 = make_float4(1.f, 0.f, 0.f, 0.f), r1 = make_float4(0.f, 1.f, 0.f, 0.f), r2 = make_float4(0.f, 0.f, 1.f, 0.f);
            float4 r3 = make_float4(1.f, 1.f, 1.f, 0.f), r4 = make_float4(__uint_as_float(0xFFFFFFFFu), 0.f, 0.f, 0.f);
            float4 r5 = r0, r6 = r1, r7 = r2;
            if (e.idx != 0xFFFFFFFFu) {
                {   // the hit-distance inputs in the checker's operation order (oracle_order_hit_t)
                    f3 t0, t1, t2;
                    rn_rotT(e.q.x, e.q.y, e.q.z, e.q.w, t0, t1, t2);
                    r5 = make_float4(t0.x, t0.y, t0.z, 1.f / e.s.x);
                    r6 = make_float4(t1.x, t1.y, t1.z, 1.f / e.s.y);
                    r7 = make_float4(t2.x, t2.y, t2.z, 1.f / e.s.z);
                }
                const m3 rt = quat_wxyz_to_rotT(e.q.x, e.q.y, e.q.z, e.q.w);
                const float ix = __builtin_amdgcn_rcpf(e.s.x), iy = __builtin_amdgcn_rcpf(e.s.y), iz = __builtin_amdgcn_rcpf(e.s.z);
                r0 = make_float4(rt.r0.x * ix, rt.r0.y * ix, rt.r0.z * ix, e.a.x);
                r1 = make_float4(rt.r1.x * iy, rt.r1.y * iy, rt.r1.z * iy, e.a.y);
                r2 = make_float4(rt.r2.x * iz, rt.r2.y * iz, rt.r2.z * iz, e.a.z);
                r3 = make_float4(e.s.x, e.s.y, e.s.z, e.a.w);
                r4.x = __uint_as_float(e.idx);
                // accept test on grayDist itself, as in the unsorted sweeps (stage_entry)
                const float need = fmaxf(P.min_response, P.min_alpha / e.a.w);
                r4.y = (P.max_alpha > P.min_alpha && e.a.w > 0.f) ? gray_limit_rt(P.degree, need) : 0.f;
            }
            float4* rec = &s_rec[lane * kQ];
            rec[0] = r0; rec[1] = r1; rec[2] = r2; rec[3] = r3; rec[4] = r4; rec[5] = r5; rec[6] = r6; rec[7] = r7;
        }
        __syncthreads();
        const int n = (int)min(64u, range.y - b);
        for (int j = 0; j < n; ++j) {
            if (!__any(alive)) break;
            const float4* rec = &s_rec[j * kQ];
            const uint32_t idx = __float_as_uint(rec[4].x);
            if (idx == 0xFFFFFFFFu) break;   // padding closes the list (gutKBufferRenderer.cuh:312-315)
            bool pop = false;
            float pop_t = 0.f, pop_a = 0.f;
            uint32_t pop_i = 0u;
            if (alive) {
                const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
                const f3 dl = ray.o - mk3(q0.w, q1.w, q2.w);
                const f3 gro = mk3(dot(mk3(q0.x, q0.y, q0.z), dl), dot(mk3(q1.x, q1.y, q1.z), dl), dot(mk3(q2.x, q2.y, q2.z), dl));
                const f3 grdu = mk3(dot(mk3(q0.x, q0.y, q0.z), ray.d), dot(mk3(q1.x, q1.y, q1.z), ray.d), dot(mk3(q2.x, q2.y, q2.z), ray.d));
                const float l2 = dot(grdu, grdu);
                const f3 gc = cross(grdu, gro);
                const float cc = dot(gc, gc);
                if (cc < rec[4].y * l2) {   // response > min_response && alpha > min_alpha
                    const float il2 = __builtin_amdgcn_rcpf(l2);
                    const float resp = response_rt(P.degree, cc * il2);
                    const float alpha = fminf(P.max_alpha, resp * q3.w);
                    // hit distance |S n (n.-u)| (gaussianParticles.slang:181-190), bits = the checker's: it is the k-buffer's sort key
                    const float4 q5 = rec[5], q6 = rec[6], q7 = rec[7];
                    const float hitT = oracle_order_hit_t(ray, mk3(q0.w, q1.w, q2.w), mk3(q3.x, q3.y, q3.z), mk3(q5.x, q5.y, q5.z), mk3(q6.x, q6.y, q6.z),
                                                          mk3(q7.x, q7.y, q7.z), mk3(q5.w, q6.w, q7.w));
                    if ((hitT > ray.tmin) && (hitT < ray.tmax)) {
                        if (kb.num == K) {   // full: the nearest pending hit is composited (below), the new one takes its slot
                            pop = true; pop_t = kb.hitT[0]; pop_a = kb.alpha[0]; pop_i = kb.idx[0];
                            kb.hitT[0] = -1.f;
                        } else {
                            kb.num++;
                        }
                        kb.insert(hitT, alpha, idx);
                    }
                }
            }
            if (BWD) {
                if (__any(pop)) {   // wave-level: gradients of lanes that popped the same particle are summed before the atomics
                    float terms[16];
                    const bool have = pop && k_bwd_terms(P, ray, density12, rgb, pop_t, pop_a, pop_i, bs, alive, terms);
                    if (kLdsFlush) k_bwd_flush_lds(have, pop_i, terms, lane, s_kterms, g_density12, g_rgb);
                    else k_bwd_flush(have, pop_i, terms, lane, g_density12, g_rgb);
                }
            } else if (pop) {
                k_process_fwd(P, rgb, pop_t, pop_a, pop_i, fs, alive);
            }
        }
        __syncthreads();
    }
    // drain what is left, nearest first (:343-351).  The buffer is ascending with the empty slots (hitT = -1) in front: K steps
    // that each take slot 0 and shift the rest down visit the pending hits in order; one copy of the per-hit code instead of K
    // (the gradient terms alone are ~600 instructions).
#pragma unroll 1
    for (int step = 0; step < K; ++step) {
        const float t0 = kb.hitT[0], a0 = kb.alpha[0];
        const uint32_t i0 = kb.idx[0];
#pragma unroll
        for (int i = 0; i + 1 < K; ++i) { kb.hitT[i] = kb.hitT[i + 1]; kb.alpha[i] = kb.alpha[i + 1]; kb.idx[i] = kb.idx[i + 1]; }
        kb.hitT[K - 1] = -1.f;
        const bool act = alive && (t0 >= 0.f);
        if (BWD) {
            if (__any(act)) {
                float terms[16];
                const bool have = act && k_bwd_terms(P, ray, density12, rgb, t0, a0, i0, bs, alive, terms);
                if (kLdsFlush) k_bwd_flush_lds(have, i0, terms, lane, s_kterms, g_density12, g_rgb);
                else k_bwd_flush(have, i0, terms, lane, g_density12, g_rgb);
            }
        } else if (act) {
            k_process_fwd(P, rgb, t0, a0, i0, fs, alive);
        }
    }
    if (!BWD && ray.inside) {
        const size_t opix = (size_t)py * P.W + px;
        const float4 o = ray.valid ? make_float4(fs.Cr, fs.Cg, fs.Cb, 1.f - fs.T) : make_float4(0.f, 0.f, 0.f, 0.f);
        store_fd(P, out_fd, opix, o);
        write_split_outputs(P, opix, o);
        out_dist[opix] = ray.valid ? fs.D : 1e6f;
        if (P.hitcounts) out_cnt[opix] = ray.valid ? fs.cnt : 0.f;
    }
}

// The forward fits three waves per SIMD when told to (1.96 vs 2.5 ms at K = 16); the backward needs its 231 registers
// (8.0 ms at two waves, 9.6 at three, 13.3 at four).
#ifndef GRUT_K_FWD_WAVES
#define GRUT_K_FWD_WAVES 3   // (measured again in round 5 with the exact-order hit distance: forward 3.07 / 2.61 / 3.08 ms at 2 / 3 / 4 waves)
#endif
template <int K>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GRUT_K_FWD_WAVES))) void gut_render_k_fwd_kernel(
    GutParams P, const uint2* __restrict__ ranges, EntryLists lists, const float4* __restrict__ density12, const float* __restrict__ rgb,
    const float* __restrict__ ray_o, const float* __restrict__ ray_d, float4* __restrict__ out_fd, float* __restrict__ out_dist,
    float* __restrict__ out_cnt) {
    gut_render_k_body<K, false>(P, ranges, lists, density12, rgb, ray_o, ray_d, out_fd, out_dist, out_cnt, nullptr, nullptr, nullptr, nullptr);
}
// K = 16 with the LDS flush: 209 VGPRs where the DPP reduce-scatter needed 231 - held to 168 (116 B of scratch) the kernel runs three waves
// per SIMD: backward 8.59 -> 7.60 ms (A/B on one box; the LDS flush at two waves: 8.98, the DPP flush at three: 9.6 in round 2).  K = 8: 177 ->
// 161 VGPRs, three waves without scratch, backward 6.53 -> 6.06 ms.
#ifndef GRUT_K_BWD_WAVES
#define GRUT_K_BWD_WAVES 3
#endif
template <int K>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu((K >= GRUT_K_FLUSH_MINK && GRUT_K_FLUSH_LDS) ? GRUT_K_BWD_WAVES : 1, (K >= GRUT_K_FLUSH_MINK && GRUT_K_FLUSH_LDS) ? GRUT_K_BWD_WAVES : 8)))
void gut_render_k_bwd_kernel(GutParams P, const uint2* __restrict__ ranges, EntryLists lists,
                                                              const float4* __restrict__ density12, const float* __restrict__ rgb,
                                                              const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                              float4* __restrict__ fd, float* __restrict__ dist, const float4* __restrict__ g_fd,
                                                              const float* __restrict__ g_dist, float* __restrict__ g_density12,
                                                              float* __restrict__ g_rgb) {
    gut_render_k_body<K, true>(P, ranges, lists, density12, rgb, ray_o, ray_d, fd, dist, nullptr, g_fd, g_dist, g_density12, g_rgb);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static uint32_t half_grid(const GutParams& P) {
    const uint32_t tiles = (uint32_t)(P.gx * P.gy);
    return ((tiles + 7u) & ~7u) * 2u;
}
// tile-first segments + one task set per segment boundary, padded to the 8-XCD interleave
static uint32_t segment_grid(const GutParams& P, uint32_t num_boundaries) {
    const uint32_t tiles = (uint32_t)(P.gx * P.gy);
    const uint32_t vtiles = ((tiles + 7u) & ~7u) + ((num_boundaries + 7u) & ~7u);
    return vtiles * 2u;
}

#define GRUT_DISPATCH_DEGREE(DEG, ...)                         \
    switch (DEG) {                                             \
    case 0: { constexpr int D_ = 0; __VA_ARGS__; } break;      \
    case 1: { constexpr int D_ = 1; __VA_ARGS__; } break;      \
    case 3: { constexpr int D_ = 3; __VA_ARGS__; } break;      \
    case 4: { constexpr int D_ = 4; __VA_ARGS__; } break;      \
    case 5: { constexpr int D_ = 5; __VA_ARGS__; } break;      \
    case 8: { constexpr int D_ = 8; __VA_ARGS__; } break;      \
    default: { constexpr int D_ = 2; __VA_ARGS__; } break;     \
    }

#if GRUT_RENDER_PART != 1   // everything but the sorted hit buffer with SH radiance
void launch_render_fwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const uint32_t* pos_particle,
                       const float* density12, const float* rgb, const float* ray_o, const float* ray_d, float* out_fd, float* out_dist,
                       float* out_cnt, const GutCheckpoints& ck, bool write_checkpoints) {
    const EntryLists lists = entry_lists(P, sorted_pos, pos_particle);
    if (P.work && P.degree == 2 && write_checkpoints) {   // instrumented frame (gut_profile_enable level 2): the counting build of the default kernel
        hipLaunchKernelGGL((gut_render_fwd_kernel<2, true, true>), dim3(half_grid(P)), dim3(64), 0, s, P, reinterpret_cast<const uint2*>(ranges), lists,
                           reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d, reinterpret_cast<float4*>(out_fd), out_dist, out_cnt, ck);
        return;
    }
    // when every wave of the launch is resident at once (<= 5 half-tile waves per SIMD x 1024 SIMDs) the longest wave is the kernel: quarter tiles
    const char* quarter_str = getenv("GRUT_FWD_QUARTER");   // development / test switch (read per call): 0 / 1 force, default by size
    const int quarter_env = quarter_str ? atoi(quarter_str) : -1;
    // (5 waves x 4 SIMDs per compute unit, from the device's own CU count: 5120 on the 256 CUs of an MI355X)
    static const uint32_t resident_limit = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 5120u;
        return 20u * (uint32_t)prop.multiProcessorCount;
    }();
    const bool quarter = quarter_env >= 0 ? quarter_env != 0 : half_grid(P) <= resident_limit;
    if (quarter && !(P.work && P.degree == 2 && write_checkpoints)) {
        const dim3 grid(half_grid(P) * 2u);
        if (write_checkpoints) {
            GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_fwd_quarter_kernel<D_, true>), grid, dim3(64), 0, s, P, reinterpret_cast<const uint2*>(ranges),
                                                              lists, reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d,
                                                              reinterpret_cast<float4*>(out_fd), out_dist, out_cnt, ck));
        } else {
            GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_fwd_quarter_kernel<D_, false>), grid, dim3(64), 0, s, P, reinterpret_cast<const uint2*>(ranges),
                                                              lists, reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d,
                                                              reinterpret_cast<float4*>(out_fd), out_dist, out_cnt, ck));
        }
        return;
    }
    if (write_checkpoints) {
        GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_fwd_kernel<D_, true>), dim3(half_grid(P)), dim3(64), 0, s, P,
                                                          reinterpret_cast<const uint2*>(ranges), lists,
                                                          reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d,
                                                          reinterpret_cast<float4*>(out_fd), out_dist, out_cnt, ck));
    } else {
        GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_fwd_kernel<D_, false>), dim3(half_grid(P)), dim3(64), 0, s, P,
                                                          reinterpret_cast<const uint2*>(ranges), lists,
                                                          reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d,
                                                          reinterpret_cast<float4*>(out_fd), out_dist, out_cnt, ck));
    }
}
void launch_render_bwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const float* density12,
                       const float* rgb, const float* ray_o, const float* ray_d, const float* fd, const GutGradIn& g_fd, const float* dist,
                       const float* g_dist, const GutGradSlots& slots, const GutCheckpoints& ck) {
    const dim3 grid(segment_grid(P, ck.num_boundaries));
    const EntryLists lists = entry_lists(P, sorted_pos, slots.pos_particle);
    if (P.work && P.degree == 2 && !g_dist) {   // instrumented frame: the counting build of the training-path kernel
        hipLaunchKernelGGL((gut_render_bwd_kernel<2, false, true>), grid, dim3(64), 0, s, P, reinterpret_cast<const uint2*>(ranges), lists,
                           reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d, reinterpret_cast<const float4*>(fd), g_fd, dist, g_dist, slots, ck);
        return;
    }
    if (g_dist) {
        GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_bwd_kernel<D_, true>), grid, dim3(64), 0, s, P,
                                                          reinterpret_cast<const uint2*>(ranges), lists,
                                                          reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d,
                                                          reinterpret_cast<const float4*>(fd), g_fd, dist,
                                                          g_dist, slots, ck));
    } else {  // no depth gradient flows in: the hit-distance terms vanish identically
        GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_bwd_kernel<D_, false>), grid, dim3(64), 0, s, P,
                                                          reinterpret_cast<const uint2*>(ranges), lists,
                                                          reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d,
                                                          reinterpret_cast<const float4*>(fd), g_fd, dist,
                                                          g_dist, slots, ck));
    }
}

// ---------------------------------------------------------------------------------------------
// Neural harmonic features, forward (model.feature_type = nht; gutKBufferRenderer.cuh:199-225, :228-352 with PerRayParticleFeatures):
// the unsorted tile loop, but a hit's features are not a per-particle colour: they are interpolated at the hit's CANONICAL INTERSECTION
// (the point of the ray closest to the particle centre, in the particle's scaled frame; gaussianParticles.slang:181-190) from the four
// feature vectors at the vertices of the canonical tetrahedron (neuralHarmonicFeaturesParticle.slang:47-66, :117-127), passed through
// the activation (:146-196) and integrated with the hit's weight into ray_dim accumulators per pixel (:198-211).  One pixel per lane,
// one wave per 16x4 strip (the k-buffer kernels' layout); the entry's K feature floats are wave-uniform and read through the scalar
// cache.  First version: correctness first, ray_dim <= 32.
// ---------------------------------------------------------------------------------------------
constexpr int kNhtMaxRay = 32, kNhtMaxIpd = 16;
// sin / cos of the activation on the hardware's v_sin_f32 / v_cos_f32 (argument in revolutions; absolute error ~1e-6 for the |angle| < 256
// this model produces — features are initialised in [-pi/2, pi/2] and multiplied by at most the frequency index): the library sinf / cosf
// cost ~40 instructions each, 48 of them per hit and pixel made up five sixths of the first version's forward
__device__ __forceinline__ float nht_sin(float x) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915494309189535f)); }
__device__ __forceinline__ float nht_cos(float x) { return __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(x * 0.15915494309189535f)); }
__global__ __launch_bounds__(64) void gut_render_nht_fwd_kernel(GutParams P, const uint2* __restrict__ ranges, EntryLists lists,
                                                                const float4* __restrict__ density12, const float* __restrict__ features,
                                                                const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                                float* __restrict__ out_fd, float* __restrict__ out_dist, float* __restrict__ out_cnt) {
    __shared__ float4 s_rec[64 * 5];
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t tile = ((slot >> 2) << 3) + xcd, strip = slot & 3u;
    if (tile >= (uint32_t)(P.gx * P.gy)) return;
    const int lane = threadIdx.x;
    const int px = (int)(tile % P.gx) * 16 + (lane & 15);
    const int py = (int)(tile / P.gx) * 16 + (int)strip * 4 + (lane >> 4);
    const Ray ray = init_ray(P, ray_o, ray_d, px, py);
    bool alive = ray.valid;
    float T = 1.f, D = 0.f, cnt = 0.f;
    float acc[kNhtMaxRay];
#pragma unroll
    for (int i = 0; i < kNhtMaxRay; ++i) acc[i] = 0.f;
    const int ipd = P.nht_ipd, nf = P.nht_nf, nr = P.nht_ray_dim;
    // the canonical tetrahedron's Cramer terms (constants of the model)
    const float edge = 4.898979485566356f, face_h = 4.242640687119285f, face_in = 1.4142135623730951f;
    const f3 v0 = mk3(0.5f * edge, -face_in, -1.f), v1 = mk3(-0.5f * edge, -face_in, -1.f), v2 = mk3(0.f, face_h - face_in, -1.f), v3 = mk3(0.f, 0.f, 3.f);
    const f3 e1 = v1 - v0, e2 = v2 - v0, e3 = v3 - v0;
    const f3 c23 = cross(e2, e3);
    const float inv_det = 1.f / dot(e1, c23);
    const uint2 range = ranges[tile];
    for (uint32_t b = range.x; b < range.y; b += 64) {
        if (!__any(alive)) break;
        {   // stage up to 64 entries (as gut_render_k_body)
            const RawEntry e = load_entry<false>(b + lane, range.y, lists, density12, nullptr);
            float4 r0 = make_float4(1.f, 0.f, 0.f, 0.f), r1 = make_float4(0.f, 1.f, 0.f, 0.f), r2 = make_float4(0.f, 0.f, 1.f, 0.f);
            float4 r3 = make_float4(1.f, 1.f, 1.f, 0.f), r4 = make_float4(__uint_as_float(0xFFFFFFFFu), 0.f, 0.f, 0.f);
            if (e.idx != 0xFFFFFFFFu) {
                const m3 rt = quat_wxyz_to_rotT(e.q.x, e.q.y, e.q.z, e.q.w);
                const float ix = 1.f / e.s.x, iy = 1.f / e.s.y, iz = 1.f / e.s.z;
                r0 = make_float4(rt.r0.x * ix, rt.r0.y * ix, rt.r0.z * ix, e.a.x);
                r1 = make_float4(rt.r1.x * iy, rt.r1.y * iy, rt.r1.z * iy, e.a.y);
                r2 = make_float4(rt.r2.x * iz, rt.r2.y * iz, rt.r2.z * iz, e.a.z);
                r3 = make_float4(e.s.x, e.s.y, e.s.z, e.a.w);
                r4.x = __uint_as_float(e.idx);
                const float need = fmaxf(P.min_response, P.min_alpha / e.a.w);
                r4.y = (P.max_alpha > P.min_alpha && e.a.w > 0.f) ? gray_limit_rt(P.degree, need) : 0.f;
            }
            float4* rec = &s_rec[lane * 5];
            rec[0] = r0; rec[1] = r1; rec[2] = r2; rec[3] = r3; rec[4] = r4;
        }
        __syncthreads();
        const int n = (int)min(64u, range.y - b);
        for (int j = 0; j < n; ++j) {
            if (!__any(alive)) break;
            const float4* rec = &s_rec[j * 5];
            const uint32_t idx = __float_as_uint(rec[4].x);
            if (idx == 0xFFFFFFFFu) break;   // padding closes the list (gutKBufferRenderer.cuh:312-315)
            bool hit = false;
            float w = 0.f;
            f3 Pc = mk3(0.f, 0.f, 0.f);
            if (alive) {
                const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
                const f3 dl = ray.o - mk3(q0.w, q1.w, q2.w);
                const f3 gro = mk3(dot(mk3(q0.x, q0.y, q0.z), dl), dot(mk3(q1.x, q1.y, q1.z), dl), dot(mk3(q2.x, q2.y, q2.z), dl));
                const f3 grdu = mk3(dot(mk3(q0.x, q0.y, q0.z), ray.d), dot(mk3(q1.x, q1.y, q1.z), ray.d), dot(mk3(q2.x, q2.y, q2.z), ray.d));
                const float l2 = dot(grdu, grdu);
                const f3 gc = cross(grdu, gro);
                const float cc = dot(gc, gc);
                if (cc < rec[4].y * l2) {   // response > min_response && alpha > min_alpha
                    const float il2 = 1.f / l2;
                    const float resp = response_rt(P.degree, cc * il2);
                    const float alpha = fminf(P.max_alpha, resp * q3.w);
                    // canonical intersection gro + grd (grd . -gro) = gro - grdu (grdu . gro) / |grdu|^2; hit distance |S (that offset)|
                    const float along = -dot(grdu, gro) * il2;
                    const f3 cg = grdu * along;
                    const f3 sv = mk3(q3.x, q3.y, q3.z) * cg;
                    const float hitT = sqrtf(dot(sv, sv));
                    if ((hitT > ray.tmin) && (hitT < ray.tmax)) {
                        w = alpha * T;
                        D = fmaf(hitT, w, D);
                        T *= (1.f - alpha);
                        hit = w > 0.f;
                        Pc = gro + cg;
                        if (hit) cnt += 1.f;
                        if (T < P.min_transmittance) alive = false;
                    }
                }
            }
            if (!__any(hit)) continue;
            // the entry's feature vectors (wave-uniform address) and this lane's barycentric weights
            const uint32_t uidx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
            float wq[4] = {1.f, 0.f, 0.f, 0.f};
            if (P.nht_support == 1) {
                const f3 d = Pc - v0;
                wq[1] = dot(d, c23) * inv_det;
                wq[2] = dot(e1, cross(d, e3)) * inv_det;
                wq[3] = dot(e1, cross(e2, d)) * inv_det;
                wq[0] = 1.f - wq[1] - wq[2] - wq[3];
            }
            const int points = P.nht_support == 1 ? 4 : 1;
            float base[kNhtMaxIpd];
#pragma unroll
            for (int m = 0; m < kNhtMaxIpd; ++m) {
                base[m] = 0.f;
                if (m < ipd) {
                    for (int k = 0; k < points; ++k) {
                        const size_t at = (size_t)uidx * P.nht_k + (size_t)k * ipd + m;
                        const float fv = P.sph_half ? __half2float(reinterpret_cast<const __half*>(features)[at]) : features[at];
                        base[m] = k == 0 ? fv * wq[0] : fmaf(wq[k], fv, base[m]);   // (:166-176: base = f0 w0, then += wk fk)
                    }
                }
            }
            if (hit) {
#pragma unroll
                for (int i = 0; i < kNhtMaxRay; ++i) {
                    if (i < nr) {
                        float f;
                        if (P.nht_act == 0) f = base[i < kNhtMaxIpd ? i : 0];
                        else if (P.nht_act == 3) f = fmaxf(0.f, base[i < kNhtMaxIpd ? i : 0]);
                        else if (P.nht_act == 2) {
                            const int k = i / (2 * nf), rem = i - k * 2 * nf, fq = rem >> 1;
                            const float angle = base[k < kNhtMaxIpd ? k : 0] * (float)(fq + 1);
                            f = (rem & 1) ? nht_cos(angle) : nht_sin(angle);
                        } else {
                            const int k = i / nf, fq = i - k * nf;
                            f = nht_sin(base[k < kNhtMaxIpd ? k : 0] * ldexpf(1.f, fq));
                        }
                        acc[i] = fmaf(f, w, acc[i]);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (ray.inside) {
        const size_t opix = (size_t)py * P.W + px;
        const size_t stride = (size_t)nr + 1;
#pragma unroll
        for (int i = 0; i < kNhtMaxRay; ++i) {
            if (i < nr) {
                const float v = ray.valid ? acc[i] : 0.f;
                if (P.out_half) reinterpret_cast<__half*>(out_fd)[opix * stride + i] = __float2half(v);
                else out_fd[opix * stride + i] = v;
            }
        }
        const float op = ray.valid ? 1.f - T : 0.f;
        if (P.out_half) reinterpret_cast<__half*>(out_fd)[opix * stride + nr] = __float2half(op);
        else out_fd[opix * stride + nr] = op;
        out_dist[opix] = ray.valid ? D : 1e6f;
        if (P.hitcounts) out_cnt[opix] = ray.valid ? cnt : 0.f;
    }
}

// Neural harmonic features, backward (evalBackwardNoKBuffer's per-ray-features branch, gutKBufferRenderer.cuh:546-641): the forward's
// sweep again, front to back, un-blending the ray state hit by hit (the reverse mode of the lerp form, like the sorted mode's backward);
// per hit and pixel the gradient of the ray features flows through the activation and the barycentric blend into (i) the particle's
// feature rows, (ii) the canonical intersection and from there, with dL/d alpha and dL/d depth, into the particle's 11 geometric terms
// (restated and checked against float64 autograd in the oracle: orc_gut_render_nht_bwd).  All 64 pixels of the strip meet a list entry
// at the same time, so the wave sums each of the entry's words over its lanes (DPP reduce-scatter, 16 words at a time) before ONE set
// of atomics per (wave, entry) — the reference does the same with warp shuffles (shRadiativeGaussianParticles.cuh:421-441).
// First version: no checkpoints (every wave sweeps its tile's list from the start), correctness first.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void gut_render_nht_bwd_kernel(GutParams P, const uint2* __restrict__ ranges, EntryLists lists,
                                                                const float4* __restrict__ density12, const float* __restrict__ features,
                                                                const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                                const float* __restrict__ fd, const float* __restrict__ g_fd,
                                                                const float* __restrict__ dist, const float* __restrict__ g_dist,
                                                                float* __restrict__ g_density12, float* __restrict__ g_features) {
    __shared__ float4 s_rec[64 * 5];
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t tile = ((slot >> 2) << 3) + xcd, strip = slot & 3u;
    if (tile >= (uint32_t)(P.gx * P.gy)) return;
    const int lane = threadIdx.x;
    const int px = (int)(tile % P.gx) * 16 + (lane & 15);
    const int py = (int)(tile / P.gx) * 16 + (int)strip * 4 + (lane >> 4);
    const Ray ray = init_ray(P, ray_o, ray_d, px, py);
    bool alive = ray.valid;
    const int ipd = P.nht_ipd, nf = P.nht_nf, nr = P.nht_ray_dim;
    const int points = P.nht_support == 1 ? 4 : 1;
    const size_t pix = ray.valid ? (size_t)py * P.W + px : 0;
    float Cb[kNhtMaxRay], gC[kNhtMaxRay];
#pragma unroll
    for (int i = 0; i < kNhtMaxRay; ++i) {
        const bool use = alive && i < nr;
        Cb[i] = use ? (P.out_half ? __half2float(reinterpret_cast<const __half*>(fd)[pix * (nr + 1) + i]) : fd[pix * (nr + 1) + i]) : 0.f;
        gC[i] = use ? g_fd[pix * (nr + 1) + i] : 0.f;
    }
    float Tb = 1.f, gT = 0.f, Db = 0.f, gD = 0.f, T = 1.f;
    if (alive) {
        const float op = P.out_half ? __half2float(reinterpret_cast<const __half*>(fd)[pix * (nr + 1) + nr]) : fd[pix * (nr + 1) + nr];
        Tb = 1.f - op; gT = -g_fd[pix * (nr + 1) + nr];
        Db = dist[pix]; gD = g_dist ? g_dist[pix] : 0.f;
    }
    const float edge = 4.898979485566356f, face_h = 4.242640687119285f, face_in = 1.4142135623730951f;
    const f3 v0 = mk3(0.5f * edge, -face_in, -1.f), v1 = mk3(-0.5f * edge, -face_in, -1.f), v2 = mk3(0.f, face_h - face_in, -1.f), v3 = mk3(0.f, 0.f, 3.f);
    const f3 e1 = v1 - v0, e2 = v2 - v0, e3 = v3 - v0;
    const f3 c23 = cross(e2, e3);
    const float inv_det = 1.f / dot(e1, c23);
    const f3 gw1 = c23 * inv_det, gw2 = cross(e3, e1) * inv_det, gw3 = cross(e1, e2) * inv_det;
    const f3 gw0 = (gw1 + gw2 + gw3) * -1.f;
    const uint2 range = ranges[tile];
    for (uint32_t b = range.x; b < range.y; b += 64) {
        if (!__any(alive)) break;
        {
            const RawEntry e = load_entry<false>(b + lane, range.y, lists, density12, nullptr);
            float4 r0 = make_float4(1.f, 0.f, 0.f, 0.f), r1 = make_float4(0.f, 1.f, 0.f, 0.f), r2 = make_float4(0.f, 0.f, 1.f, 0.f);
            float4 r3 = make_float4(1.f, 1.f, 1.f, 0.f), r4 = make_float4(__uint_as_float(0xFFFFFFFFu), 0.f, 0.f, 0.f);
            if (e.idx != 0xFFFFFFFFu) {   // raw rows: the gradient chain needs the quaternion and the scale themselves
                r0 = e.a; r1 = e.q; r2 = make_float4(e.s.x, e.s.y, e.s.z, 0.f);
                r4.x = __uint_as_float(e.idx);
                const float need = fmaxf(P.min_response, P.min_alpha / e.a.w);
                r4.y = (P.max_alpha > P.min_alpha && e.a.w > 0.f) ? gray_limit_rt(P.degree, need) : 0.f;
            }
            float4* rec = &s_rec[lane * 5];
            rec[0] = r0; rec[1] = r1; rec[2] = r2; rec[3] = r3; rec[4] = r4;
        }
        __syncthreads();
        const int n = (int)min(64u, range.y - b);
        for (int j = 0; j < n; ++j) {
            if (!__any(alive)) break;
            const float4* rec = &s_rec[j * 5];
            const uint32_t idx = __float_as_uint(rec[4].x);
            if (idx == 0xFFFFFFFFu) break;
            const float4 a = rec[0], q = rec[1], sc = rec[2];
            bool hit = false;
            float gd[16], gbase[kNhtMaxIpd], wq[4] = {1.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) gd[k] = 0.f;
#pragma unroll
            for (int m = 0; m < kNhtMaxIpd; ++m) gbase[m] = 0.f;
            if (alive) {
                const m3 rotT = quat_wxyz_to_rotT(q.x, q.y, q.z, q.w);
                const f3 gscl = mk3(sc.x, sc.y, sc.z), giscl = mk3(1.f / sc.x, 1.f / sc.y, 1.f / sc.z);
                const f3 gposc = ray.o - mk3(a.x, a.y, a.z);
                const f3 gposcr = mul_rows(rotT, gposc);
                const f3 gro = giscl * gposcr;
                const f3 rdr = mul_rows(rotT, ray.d);
                const f3 grdu = giscl * rdr;
                const float l2 = dot(grdu, grdu);
                const float il = 1.f / sqrtf(l2);
                const f3 grd = grdu * il;
                const f3 gcrod = cross(grd, gro);
                const float gray = dot(gcrod, gcrod);
                if (gray < rec[4].y) {
                    const float gres = response_rt(P.degree, gray);
                    const float alpha = fminf(P.max_alpha, gres * a.w);
                    const float pdot = -dot(grd, gro);
                    const f3 grdd = grd * pdot;
                    const f3 Pc = gro + grdd;
                    const f3 grds = gscl * grdd;
                    const float gsq = dot(grds, grds);
                    const float hitT = sqrtf(gsq);
                    if ((hitT > ray.tmin) && (hitT < ray.tmax)) {
                        hit = alpha > 0.f;
                        if (P.nht_support == 1) {
                            const f3 d = Pc - v0;
                            wq[1] = dot(d, c23) * inv_det; wq[2] = dot(e1, cross(d, e3)) * inv_det; wq[3] = dot(e1, cross(e2, d)) * inv_det;
                            wq[0] = 1.f - wq[1] - wq[2] - wq[3];
                        }
                        float base[kNhtMaxIpd];
#pragma unroll
                        for (int m = 0; m < kNhtMaxIpd; ++m) {
                            base[m] = 0.f;
                            if (m < ipd)
                                for (int k = 0; k < points; ++k) {
                                    const size_t at = (size_t)idx * P.nht_k + (size_t)k * ipd + m;
                                    const float fv = P.sph_half ? __half2float(reinterpret_cast<const __half*>(features)[at]) : features[at];
                                    base[m] = k == 0 ? fv * wq[0] : fmaf(wq[k], fv, base[m]);
                                }
                        }
                        const float w = 1.f / (1.f - alpha);
                        float dalpha = 0.f;
#pragma unroll
                        for (int i = 0; i < kNhtMaxRay; ++i) {
                            if (i < nr) {
                                // feature i and d feature i / d base (sincos: index k*nf*2 + f*2 + {0,1}; siren: k*nf + f)
                                float f, df;
                                int kb;
                                if (P.nht_act == 0) { kb = i; f = base[kb < kNhtMaxIpd ? kb : 0]; df = 1.f; }
                                else if (P.nht_act == 3) { kb = i; const float bv = base[kb < kNhtMaxIpd ? kb : 0]; f = fmaxf(0.f, bv); df = bv > 0.f ? 1.f : 0.f; }
                                else if (P.nht_act == 2) {
                                    kb = i / (2 * nf);
                                    const int rem = i - kb * 2 * nf, fq = rem >> 1;
                                    const float fr = (float)(fq + 1), ang = base[kb < kNhtMaxIpd ? kb : 0] * fr;
                                    const float sn = nht_sin(ang), cs = nht_cos(ang);
                                    f = (rem & 1) ? cs : sn; df = (rem & 1) ? -fr * sn : fr * cs;
                                } else {
                                    kb = i / nf;
                                    const float fr = ldexpf(1.f, i - kb * nf), ang = base[kb < kNhtMaxIpd ? kb : 0] * fr;
                                    f = nht_sin(ang); df = fr * nht_cos(ang);
                                }
                                if (hit) {
                                    Cb[i] = (Cb[i] - f * alpha) * w;
                                    dalpha = fmaf(f - Cb[i], gC[i], dalpha);
                                    const float gf = alpha * gC[i];
                                    gC[i] *= (1.f - alpha);
#pragma unroll
                                    for (int m = 0; m < kNhtMaxIpd; ++m)
                                        if (m == kb) gbase[m] = fmaf(df, gf, gbase[m]);
                                }
                            }
                        }
                        // blend backward: the canonical position's gradient (the feature rows' gradient is wq[k] * gbase, summed over the wave below)
                        f3 dP = mk3(0.f, 0.f, 0.f);
                        if (P.nht_support == 1 && hit) {
                            float dw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int m = 0; m < kNhtMaxIpd; ++m)
                                if (m < ipd)
                                    for (int k = 0; k < 4; ++k) {
                                        const size_t at = (size_t)idx * P.nht_k + (size_t)k * ipd + m;
                                        const float fv = P.sph_half ? __half2float(reinterpret_cast<const __half*>(features)[at]) : features[at];
                                        dw[k] = fmaf(fv, gbase[m], dw[k]);
                                    }
                            dP = gw0 * dw[0] + gw1 * dw[1] + gw2 * dw[2] + gw3 * dw[3];
                        }
                        // density: T_out = T_in (1 - alpha), D_front = lerp(D_behind, depth, alpha)
                        Tb *= w;
                        Db = (Db - hitT * alpha) * w;
                        dalpha += (hitT - Db) * gD - Tb * gT;
                        const float ddepth = alpha * gD;
                        gD *= (1.f - alpha);
                        gT *= (1.f - alpha);
                        if (hit) {
                            float dres = 0.f, ddens = 0.f;
                            if (gres * a.w < P.max_alpha) { dres = a.w * dalpha; ddens = gres * dalpha; }
                            const float grayGrd = response_grd_rt(P.degree, gray, gres, dres);
                            const f3 grdsGrd = gsq > 0.f ? grds * (ddepth / hitT) : mk3(0.f, 0.f, 0.f);
                            const f3 gsclHit = grdd * grdsGrd;
                            const float sdot = dot(grdsGrd * gscl, grd);
                            const float gdP = dot(grd, dP);
                            const f3 grdHit = gscl * grdsGrd * pdot - gro * sdot + dP * pdot - gro * gdP;
                            const f3 groHit = grd * (-sdot) + dP - grd * gdP;
                            const f3 gcrodGrd = gcrod * (2.f * grayGrd);
                            const f3 grdGrd = mk3(gcrodGrd.z * gro.y - gcrodGrd.y * gro.z, gcrodGrd.x * gro.z - gcrodGrd.z * gro.x, gcrodGrd.y * gro.x - gcrodGrd.x * gro.y);
                            const f3 groGrd = mk3(gcrodGrd.y * grd.z - gcrodGrd.z * grd.y, gcrodGrd.z * grd.x - gcrodGrd.x * grd.z, gcrodGrd.x * grd.y - gcrodGrd.y * grd.x);
                            const f3 groTot = groGrd + groHit;
                            const f3 is2 = giscl * giscl;
                            const f3 gsclGro = mk3(-gposcr.x * is2.x, -gposcr.y * is2.y, -gposcr.z * is2.z) * groTot;
                            const f3 gposcrGrd = giscl * groTot;
                            const f3 gposcGrd = mul_cols(rotT, gposcrGrd);
                            const f3 dn = grdGrd + grdHit;
                            const f3 grduGrd = dn * il - grdu * (il * il * il * dot(dn, grdu));   // normalize backward
                            const f3 sclGrd = gsclHit + gsclGro + mk3(-rdr.x * is2.x, -rdr.y * is2.y, -rdr.z * is2.z) * grduGrd;
                            const float4 gq1 = quat_outer_contract(gposcrGrd, gposc, q), gq2 = quat_outer_contract(giscl * grduGrd, ray.d, q);
                            gd[0] = -gposcGrd.x; gd[1] = -gposcGrd.y; gd[2] = -gposcGrd.z; gd[3] = ddens;
                            gd[4] = gq1.x + gq2.x; gd[5] = gq1.y + gq2.y; gd[6] = gq1.z + gq2.z; gd[7] = gq1.w + gq2.w;
                            gd[8] = sclGrd.x; gd[9] = sclGrd.y; gd[10] = sclGrd.z;
                        }
                        T *= (1.f - alpha);
                        if (T < P.min_transmittance) alive = false;
                    }
                }
            }
            if (!__any(hit)) continue;
            // one set of atomics per (wave, entry): 11 geometric words, then the feature rows point by point
            const float tot = wave_reduce_scatter16(gd, lane);
            if (lane < 11) atomicAdd(g_density12 + 12 * (size_t)idx + lane, tot);
            for (int k = 0; k < points; ++k) {
                float gfk[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) gfk[m] = (hit && m < ipd) ? wq[k] * gbase[m < kNhtMaxIpd ? m : 0] : 0.f;
                const float t2 = wave_reduce_scatter16(gfk, lane);
                if (lane < ipd) atomicAdd(g_features + (size_t)idx * P.nht_k + (size_t)k * ipd + lane, t2);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Neural harmonic features behind the SORTED hit buffer (render.splat.k_buffer_size > 0 with model.feature_type = nht; round 6):
// gutKBufferRenderer.cuh:273-352 (evalKBuffer) with Params::PerRayParticleFeatures - processHitParticle :158-225 integrates, per popped hit,
// the features interpolated at the hit's canonical intersection.  The tile loop, the hit test, the hit distance in the checker's operation
// order (it is the buffer's sort key) and the buffer itself are gut_render_k_body's; a popped hit is a (particle, alpha, distance) triple,
// everything else the feature integration needs - the canonical intersection, the response for the gradient chain - is a function of
// (ray, particle) and is evaluated again at the pop from the particle's own rows (the reference keeps the intersection in the buffer: the
// same value).  The gradients leave per lane and hit with atomics, like the reference's (featuresIntegrateBwdToBuffer /
// densityProcessHitBwdToBuffer); the per-hit arithmetic is the unsorted strip kernels' above, hit by hit.
// ---------------------------------------------------------------------------------------------
struct NhtTetra4 { f3 v0, e1, e2, e3, c23, gw0, gw1, gw2, gw3; float inv_det; };
__device__ __forceinline__ NhtTetra4 nht_tetra4() {
    NhtTetra4 t;
    const float edge = 4.898979485566356f, face_h = 4.242640687119285f, face_in = 1.4142135623730951f;
    const f3 v1 = mk3(-0.5f * edge, -face_in, -1.f), v2 = mk3(0.f, face_h - face_in, -1.f), v3 = mk3(0.f, 0.f, 3.f);
    t.v0 = mk3(0.5f * edge, -face_in, -1.f);
    t.e1 = v1 - t.v0; t.e2 = v2 - t.v0; t.e3 = v3 - t.v0;
    t.c23 = cross(t.e2, t.e3);
    t.inv_det = 1.f / dot(t.e1, t.c23);
    t.gw1 = t.c23 * t.inv_det; t.gw2 = cross(t.e3, t.e1) * t.inv_det; t.gw3 = cross(t.e1, t.e2) * t.inv_det;
    t.gw0 = (t.gw1 + t.gw2 + t.gw3) * -1.f;
    return t;
}
__device__ __forceinline__ float nht_feature_word(const GutParams& P, const float* __restrict__ features, uint32_t idx, int word) {
    const size_t at = (size_t)idx * P.nht_k + word;
    return P.sph_half ? __half2float(reinterpret_cast<const __half*>(features)[at]) : features[at];
}
// a popped hit's particle is the lane's own: its feature row is gathered per lane.  The default model's row (48 floats = 4 vertices x 12,
// fp32) comes as twelve 16-byte loads instead of 48 single words (the first version's forward spent most of its time issuing them)
__host__ __device__ __forceinline__ bool nht_row48(const GutParams& P) { return P.nht_k == 48 && P.nht_ipd == 12 && P.nht_support == 1 && !P.sph_half; }
__device__ __forceinline__ void nht_row_blend(const GutParams& P, const float* __restrict__ features, uint32_t idx, const float (&wq)[4], float (&base)[kNhtMaxIpd]) {
    const int points = P.nht_support == 1 ? 4 : 1, ipd = P.nht_ipd;
#pragma unroll
    for (int m = 0; m < kNhtMaxIpd; ++m) base[m] = 0.f;
    if (nht_row48(P)) {
        const float4* src = reinterpret_cast<const float4*>(features + (size_t)idx * 48);
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const float4 v = src[q];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                constexpr int kIpd = 12;
                const int w = 4 * q + c, k = w / kIpd, m = w - k * kIpd;
                base[m] = k == 0 ? e[c] * wq[0] : fmaf(wq[k], e[c], base[m]);   // (base = f0 w0, then += wk fk: the strip kernels' order)
            }
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < kNhtMaxIpd; ++m)
        if (m < ipd)
            for (int k = 0; k < points; ++k) {
                const float fv = nht_feature_word(P, features, idx, k * ipd + m);
                base[m] = k == 0 ? fv * wq[0] : fmaf(wq[k], fv, base[m]);
            }
}
// dw[k] = sum_m F[k][m] gbase[m]: the barycentric weights' gradients
__device__ __forceinline__ void nht_row_dots(const GutParams& P, const float* __restrict__ features, uint32_t idx, const float (&gbase)[kNhtMaxIpd], float (&dw)[4]) {
    const int points = P.nht_support == 1 ? 4 : 1, ipd = P.nht_ipd;
    dw[0] = dw[1] = dw[2] = dw[3] = 0.f;
    if (nht_row48(P)) {
        const float4* src = reinterpret_cast<const float4*>(features + (size_t)idx * 48);
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const float4 v = src[q];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                constexpr int kIpd = 12;
                const int w = 4 * q + c, k = w / kIpd, m = w - k * kIpd;
                dw[k] = fmaf(e[c], gbase[m], dw[k]);
            }
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < kNhtMaxIpd; ++m)
        if (m < ipd)
            for (int k = 0; k < points; ++k) dw[k] = fmaf(nht_feature_word(P, features, idx, k * ipd + m), gbase[m], dw[k]);
}
// feature i of the activation and its derivative with respect to its base feature kb (the unsorted strip kernels' rules)
__device__ __forceinline__ void nht_activation_rt(const GutParams& P, const float (&base)[kNhtMaxIpd], int i, float& f, float& df, int& kb) {
    const int nf = P.nht_nf;
    if (P.nht_act == 0) { kb = i; f = base[kb < kNhtMaxIpd ? kb : 0]; df = 1.f; }
    else if (P.nht_act == 3) { kb = i; const float bv = base[kb < kNhtMaxIpd ? kb : 0]; f = fmaxf(0.f, bv); df = bv > 0.f ? 1.f : 0.f; }
    else if (P.nht_act == 2) {
        kb = i / (2 * nf);
        const int rem = i - kb * 2 * nf, fq = rem >> 1;
        const float fr = (float)(fq + 1), ang = base[kb < kNhtMaxIpd ? kb : 0] * fr;
        const float sn = nht_sin(ang), cs = nht_cos(ang);
        f = (rem & 1) ? cs : sn; df = (rem & 1) ? -fr * sn : fr * cs;
    } else {
        kb = i / nf;
        const float fr = ldexpf(1.f, i - kb * nf), ang = base[kb < kNhtMaxIpd ? kb : 0] * fr;
        f = nht_sin(ang); df = fr * nht_cos(ang);
    }
}
struct NhtKFwdState { float T, D, cnt; float acc[kNhtMaxRay]; };
// FAST: the default feature model (48 = 4 x 12 fp32 floats, sincos with one frequency -> 24 ray features) as compile-time constants - the
// arrays shrink to what it uses and a base feature's sine and cosine are evaluated once for its two ray features (446 VGPRs -> see obj_resources)
template <bool FAST>
__device__ __forceinline__ void nht_k_pop_fwd(const GutParams& P, const Ray& ray, const NhtTetra4& tet, const float4* __restrict__ density12,
                                              const float* __restrict__ features, float hitT, float alpha, uint32_t idx, NhtKFwdState& s, bool& alive) {
    const float w = alpha * s.T;
    s.D = fmaf(hitT, w, s.D);
    s.T *= (1.f - alpha);
    if (w > 0.f) {
        const float4 a = density12[3 * (size_t)idx], q = density12[3 * (size_t)idx + 1], sc = density12[3 * (size_t)idx + 2];
        const m3 rt = quat_wxyz_to_rotT(q.x, q.y, q.z, q.w);
        const f3 giscl = mk3(1.f / sc.x, 1.f / sc.y, 1.f / sc.z);
        const f3 gro = giscl * mul_rows(rt, ray.o - mk3(a.x, a.y, a.z));
        const f3 grdu = giscl * mul_rows(rt, ray.d);
        const float along = -dot(grdu, gro) / dot(grdu, grdu);
        const f3 Pc = gro + grdu * along;
        float wq[4] = {1.f, 0.f, 0.f, 0.f};
        if (P.nht_support == 1) {
            const f3 d = Pc - tet.v0;
            wq[1] = dot(d, tet.c23) * tet.inv_det; wq[2] = dot(tet.e1, cross(d, tet.e3)) * tet.inv_det; wq[3] = dot(tet.e1, cross(tet.e2, d)) * tet.inv_det;
            wq[0] = 1.f - wq[1] - wq[2] - wq[3];
        }
        const int nr = P.nht_ray_dim;
        float base[kNhtMaxIpd];
        nht_row_blend(P, features, idx, wq, base);
        if (FAST) {
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                s.acc[2 * m] = fmaf(nht_sin(base[m]), w, s.acc[2 * m]);
                s.acc[2 * m + 1] = fmaf(nht_cos(base[m]), w, s.acc[2 * m + 1]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < kNhtMaxRay; ++i) {
                if (i < nr) {
                    float f, df; int kb;
                    nht_activation_rt(P, base, i, f, df, kb);
                    s.acc[i] = fmaf(f, w, s.acc[i]);
                }
            }
        }
        s.cnt += 1.f;
    }
    if (s.T < P.min_transmittance) alive = false;
}
struct NhtKBwdState { float Cb[kNhtMaxRay], gC[kNhtMaxRay], Tb, gT, Db, gD, T; };
// returns whether the hit carries gradients; gd = its 11 geometric words, wq / gbase = the factors of its feature-row words (wq[k] * gbase[m])
template <bool FAST>
__device__ __forceinline__ bool nht_k_pop_bwd(const GutParams& P, const Ray& ray, const NhtTetra4& tet, const float4* __restrict__ density12,
                                              const float* __restrict__ features, float hitT, float alpha, uint32_t idx, NhtKBwdState& s, bool& alive,
                                              float (&gd)[11], float (&wq)[4], float (&gbase)[kNhtMaxIpd]) {
    const int points = P.nht_support == 1 ? 4 : 1, ipd = P.nht_ipd, nr = P.nht_ray_dim;
    const float4 a = density12[3 * (size_t)idx], q = density12[3 * (size_t)idx + 1], sc = density12[3 * (size_t)idx + 2];
    const m3 rotT = quat_wxyz_to_rotT(q.x, q.y, q.z, q.w);
    const f3 gscl = mk3(sc.x, sc.y, sc.z), giscl = mk3(1.f / sc.x, 1.f / sc.y, 1.f / sc.z);
    const f3 gposc = ray.o - mk3(a.x, a.y, a.z);
    const f3 gposcr = mul_rows(rotT, gposc);
    const f3 gro = giscl * gposcr;
    const f3 rdr = mul_rows(rotT, ray.d);
    const f3 grdu = giscl * rdr;
    const float l2 = dot(grdu, grdu);
    const float il = 1.f / sqrtf(l2);
    const f3 grd = grdu * il;
    const f3 gcrod = cross(grd, gro);
    const float gray = dot(gcrod, gcrod);
    const float gres = response_rt(P.degree, gray);
    const float pdot = -dot(grd, gro);
    const f3 grdd = grd * pdot;
    const f3 Pc = gro + grdd;
    const f3 grds = gscl * grdd;
    const float gsq = dot(grds, grds);
    const bool hit = alpha > 0.f;
    wq[0] = 1.f; wq[1] = 0.f; wq[2] = 0.f; wq[3] = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) gd[k] = 0.f;
    if (P.nht_support == 1) {
        const f3 d = Pc - tet.v0;
        wq[1] = dot(d, tet.c23) * tet.inv_det; wq[2] = dot(tet.e1, cross(d, tet.e3)) * tet.inv_det; wq[3] = dot(tet.e1, cross(tet.e2, d)) * tet.inv_det;
        wq[0] = 1.f - wq[1] - wq[2] - wq[3];
    }
    float base[kNhtMaxIpd];
    nht_row_blend(P, features, idx, wq, base);
#pragma unroll
    for (int m = 0; m < kNhtMaxIpd; ++m) gbase[m] = 0.f;
    const float w = 1.f / (1.f - alpha);
    float dalpha = 0.f;
    if (hit && FAST) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            const float sn = nht_sin(base[m]), cs = nht_cos(base[m]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {   // ray features 2 m (sine) and 2 m + 1 (cosine), in this order like the generic loop
                const int i = 2 * m + h;
                const float f = h ? cs : sn, df = h ? -sn : cs;
                s.Cb[i] = (s.Cb[i] - f * alpha) * w;
                dalpha = fmaf(f - s.Cb[i], s.gC[i], dalpha);
                const float gf = alpha * s.gC[i];
                s.gC[i] *= (1.f - alpha);
                gbase[m] = fmaf(df, gf, gbase[m]);
            }
        }
    } else if (hit) {
#pragma unroll
        for (int i = 0; i < kNhtMaxRay; ++i) {
            if (i < nr) {
                float f, df; int kb;
                nht_activation_rt(P, base, i, f, df, kb);
                s.Cb[i] = (s.Cb[i] - f * alpha) * w;
                dalpha = fmaf(f - s.Cb[i], s.gC[i], dalpha);
                const float gf = alpha * s.gC[i];
                s.gC[i] *= (1.f - alpha);
#pragma unroll
                for (int m = 0; m < kNhtMaxIpd; ++m)
                    if (m == kb) gbase[m] = fmaf(df, gf, gbase[m]);
            }
        }
    }
    // blend backward: the canonical position (the feature rows' words wq[k] * gbase[m] leave through nht_k_bwd_flush)
    f3 dP = mk3(0.f, 0.f, 0.f);
    if (hit && P.nht_support == 1) {
        float dw[4];
        nht_row_dots(P, features, idx, gbase, dw);
        dP = tet.gw0 * dw[0] + tet.gw1 * dw[1] + tet.gw2 * dw[2] + tet.gw3 * dw[3];
    }
    // density: T_out = T_in (1 - alpha), D_front = lerp(D_behind, depth, alpha)
    s.Tb *= w;
    s.Db = (s.Db - hitT * alpha) * w;
    dalpha += (hitT - s.Db) * s.gD - s.Tb * s.gT;
    const float ddepth = alpha * s.gD;
    s.gD *= (1.f - alpha);
    s.gT *= (1.f - alpha);
    if (hit) {
        float dres = 0.f, ddens = 0.f;
        if (gres * a.w < P.max_alpha) { dres = a.w * dalpha; ddens = gres * dalpha; }
        const float grayGrd = response_grd_rt(P.degree, gray, gres, dres);
        const f3 grdsGrd = gsq > 0.f ? grds * (ddepth / hitT) : mk3(0.f, 0.f, 0.f);
        const f3 gsclHit = grdd * grdsGrd;
        const float sdot = dot(grdsGrd * gscl, grd);
        const float gdP = dot(grd, dP);
        const f3 grdHit = gscl * grdsGrd * pdot - gro * sdot + dP * pdot - gro * gdP;
        const f3 groHit = grd * (-sdot) + dP - grd * gdP;
        const f3 gcrodGrd = gcrod * (2.f * grayGrd);
        const f3 grdGrd = mk3(gcrodGrd.z * gro.y - gcrodGrd.y * gro.z, gcrodGrd.x * gro.z - gcrodGrd.z * gro.x, gcrodGrd.y * gro.x - gcrodGrd.x * gro.y);
        const f3 groGrd = mk3(gcrodGrd.y * grd.z - gcrodGrd.z * grd.y, gcrodGrd.z * grd.x - gcrodGrd.x * grd.z, gcrodGrd.x * grd.y - gcrodGrd.y * grd.x);
        const f3 groTot = groGrd + groHit;
        const f3 is2 = giscl * giscl;
        const f3 gsclGro = mk3(-gposcr.x * is2.x, -gposcr.y * is2.y, -gposcr.z * is2.z) * groTot;
        const f3 gposcrGrd = giscl * groTot;
        const f3 gposcGrd = mul_cols(rotT, gposcrGrd);
        const f3 dn = grdGrd + grdHit;
        const f3 grduGrd = dn * il - grdu * (il * il * il * dot(dn, grdu));   // normalize backward
        const f3 sclGrd = gsclHit + gsclGro + mk3(-rdr.x * is2.x, -rdr.y * is2.y, -rdr.z * is2.z) * grduGrd;
        const float4 gq1 = quat_outer_contract(gposcrGrd, gposc, q), gq2 = quat_outer_contract(giscl * grduGrd, ray.d, q);
        gd[0] = -gposcGrd.x; gd[1] = -gposcGrd.y; gd[2] = -gposcGrd.z; gd[3] = ddens;
        gd[4] = gq1.x + gq2.x; gd[5] = gq1.y + gq2.y; gd[6] = gq1.z + gq2.z; gd[7] = gq1.w + gq2.w;
        gd[8] = sclGrd.x; gd[9] = sclGrd.y; gd[10] = sclGrd.z;
    }
    s.T *= (1.f - alpha);
    if (s.T < P.min_transmittance) alive = false;
    return hit;
}
// The gradients of a step's popped hits, hit-major through LDS (k_bwd_flush_lds / the 3DGRT feature replay): every popping lane parks its 11
// geometric words, 4 barycentric weights and <= 16 base-feature gradients; per DISTINCT particle of the step the lanes ARE the words of its
// rows - feature word (k, m) = sum over the member lanes of wq[k] * gbase[m], then the 11 geometric words - and leave as one atomic
// instruction per 64 consecutive words.  (Per-lane atomics of all 59 words, the reference's own scheme and this kernel's first version: 644 ms
// for the 1 M / 1080p backward.)
constexpr int kNhtKTermStride = 33;
template <bool FAST>
__device__ __forceinline__ void nht_k_bwd_flush(const GutParams& P, bool have, uint32_t idx, const float (&gd)[11], const float (&wq)[4],
                                                const float (&gbase)[kNhtMaxIpd], int lane, float* __restrict__ s_terms, float* __restrict__ g_density12,
                                                float* __restrict__ g_features) {
    if (have) {
        float* tw = s_terms + lane * kNhtKTermStride;
#pragma unroll
        for (int k = 0; k < 11; ++k) tw[k] = gd[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) tw[11 + k] = wq[k];
#pragma unroll
        for (int m = 0; m < (FAST ? 12 : kNhtMaxIpd); ++m) tw[15 + m] = gbase[m];
    }
    __syncthreads();   // single-wave workgroup: orders the LDS hand-off
    const int points = FAST ? 4 : (P.nht_support == 1 ? 4 : 1), ipd = FAST ? 12 : P.nht_ipd, words = points * ipd;
    unsigned long long m = __ballot(have);
    while (m) {
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t pid = (uint32_t)__builtin_amdgcn_readlane((int)idx, leader);
        const unsigned long long same = __ballot(have && idx == pid);
        m &= ~same;
        for (int chunk = 0; chunk < words; chunk += 64) {
            const int word = chunk + lane;
            const int kq = word < words ? word / ipd : 0, nq = word < words ? word - kq * ipd : 0;
            float v = 0.f;
            for (unsigned long long r = same; r; r &= r - 1) {
                const float* tw = s_terms + (__ffsll((long long)r) - 1) * kNhtKTermStride;
                v = fmaf(tw[11 + kq], tw[15 + nq], v);
            }
            if (word < words && v != 0.f) atomicAdd(g_features + (size_t)pid * P.nht_k + word, v);
        }
        float v = 0.f;
        for (unsigned long long r = same; r; r &= r - 1) v += s_terms[(__ffsll((long long)r) - 1) * kNhtKTermStride + (lane < 11 ? lane : 0)];
        if (lane < 11 && v != 0.f) atomicAdd(g_density12 + 12 * (size_t)pid + lane, v);
    }
    __syncthreads();   // the next step overwrites s_terms
}
#ifndef GRUT_NHT_K_WAVES
#define GRUT_NHT_K_WAVES 2   // the default model's instantiations are held to two waves per SIMD (256 VGPRs): forward K = 16 259 -> 256
#endif
template <int K, bool BWD, bool FAST>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FAST ? GRUT_NHT_K_WAVES : 1, 8))) void gut_render_nht_k_kernel(GutParams P, const uint2* __restrict__ ranges, EntryLists lists,
                                                              const float4* __restrict__ density12, const float* __restrict__ features,
                                                              const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                              float* __restrict__ fd /* BWD: the forward's image (read) */, float* __restrict__ dist,
                                                              float* __restrict__ out_cnt, const float* __restrict__ g_fd, const float* __restrict__ g_dist,
                                                              float* __restrict__ g_density12, float* __restrict__ g_features) {
    constexpr int kQ = 8;   // as gut_render_k_body: 0-2 M rows | pos, 3 scale | density, 4 particle | accept limit, 5-7 rows of R^T | 1 / scale
    __shared__ float4 s_rec[64 * kQ];
    __shared__ float s_kterms[BWD ? 64 * kNhtKTermStride : 1];
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t tile = ((slot >> 2) << 3) + xcd, strip = slot & 3u;
    if (tile >= (uint32_t)(P.gx * P.gy)) return;
    const int lane = threadIdx.x;
    const int px = (int)(tile % P.gx) * 16 + (lane & 15);
    const int py = (int)(tile / P.gx) * 16 + (int)strip * 4 + (lane >> 4);
    const Ray ray = init_ray(P, ray_o, ray_d, px, py);
    bool alive = ray.valid;
    const size_t pix = ray.valid ? (size_t)py * P.W + px : 0;
    const int nr = FAST ? 24 : P.nht_ray_dim;
    const NhtTetra4 tet = nht_tetra4();
    NhtKFwdState fs;
    NhtKBwdState bs;
    if (!BWD) {
        fs.T = 1.f; fs.D = 0.f; fs.cnt = 0.f;
#pragma unroll
        for (int i = 0; i < (FAST ? 24 : kNhtMaxRay); ++i) fs.acc[i] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < (FAST ? 24 : kNhtMaxRay); ++i) {
            const bool use = alive && i < nr;
            bs.Cb[i] = use ? (P.out_half ? __half2float(reinterpret_cast<const __half*>(fd)[pix * (nr + 1) + i]) : fd[pix * (nr + 1) + i]) : 0.f;
            bs.gC[i] = use ? g_fd[pix * (nr + 1) + i] : 0.f;
        }
        bs.Tb = 1.f; bs.gT = 0.f; bs.Db = 0.f; bs.gD = 0.f; bs.T = 1.f;
        if (alive) {
            const float op = P.out_half ? __half2float(reinterpret_cast<const __half*>(fd)[pix * (nr + 1) + nr]) : fd[pix * (nr + 1) + nr];
            bs.Tb = 1.f - op; bs.gT = -g_fd[pix * (nr + 1) + nr];
            bs.Db = dist[pix]; bs.gD = g_dist ? g_dist[pix] : 0.f;
        }
    }
    KBuffer<K> kb;
    kb.clear();
    const uint2 range = ranges[tile];
    for (uint32_t b = range.x; b < range.y; b += 64) {
        if (!__any(alive)) break;
        {   // stage up to 64 entries (gut_render_k_body)
            const RawEntry e = load_entry<false>(b + lane, range.y, lists, density12, nullptr);
            float4 r0 = make_float4(1.f, 0.f, 0.f, 0.f), r1 = make_float4(0.f, 1.f, 0.f, 0.f), r2 = make_float4(0.f, 0.f, 1.f, 0.f);
            float4 r3 = make_float4(1.f, 1.f, 1.f, 0.f), r4 = make_float4(__uint_as_float(0xFFFFFFFFu), 0.f, 0.f, 0.f);
            float4 r5 = r0, r6 = r1, r7 = r2;
            if (e.idx != 0xFFFFFFFFu) {
                {
                    f3 t0, t1, t2;
                    rn_rotT(e.q.x, e.q.y, e.q.z, e.q.w, t0, t1, t2);
                    r5 = make_float4(t0.x, t0.y, t0.z, 1.f / e.s.x);
                    r6 = make_float4(t1.x, t1.y, t1.z, 1.f / e.s.y);
                    r7 = make_float4(t2.x, t2.y, t2.z, 1.f / e.s.z);
                }
                const m3 rt = quat_wxyz_to_rotT(e.q.x, e.q.y, e.q.z, e.q.w);
                const float ix = __builtin_amdgcn_rcpf(e.s.x), iy = __builtin_amdgcn_rcpf(e.s.y), iz = __builtin_amdgcn_rcpf(e.s.z);
                r0 = make_float4(rt.r0.x * ix, rt.r0.y * ix, rt.r0.z * ix, e.a.x);
                r1 = make_float4(rt.r1.x * iy, rt.r1.y * iy, rt.r1.z * iy, e.a.y);
                r2 = make_float4(rt.r2.x * iz, rt.r2.y * iz, rt.r2.z * iz, e.a.z);
                r3 = make_float4(e.s.x, e.s.y, e.s.z, e.a.w);
                r4.x = __uint_as_float(e.idx);
                const float need = fmaxf(P.min_response, P.min_alpha / e.a.w);
                r4.y = (P.max_alpha > P.min_alpha && e.a.w > 0.f) ? gray_limit_rt(P.degree, need) : 0.f;
            }
            float4* rec = &s_rec[lane * kQ];
            rec[0] = r0; rec[1] = r1; rec[2] = r2; rec[3] = r3; rec[4] = r4; rec[5] = r5; rec[6] = r6; rec[7] = r7;
        }
        __syncthreads();
        const int n = (int)min(64u, range.y - b);
        for (int j = 0; j < n; ++j) {
            if (!__any(alive)) break;
            const float4* rec = &s_rec[j * kQ];
            const uint32_t idx = __float_as_uint(rec[4].x);
            if (idx == 0xFFFFFFFFu) break;
            bool pop = false;
            float pop_t = 0.f, pop_a = 0.f;
            uint32_t pop_i = 0u;
            if (alive) {
                const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
                const f3 dl = ray.o - mk3(q0.w, q1.w, q2.w);
                const f3 gro = mk3(dot(mk3(q0.x, q0.y, q0.z), dl), dot(mk3(q1.x, q1.y, q1.z), dl), dot(mk3(q2.x, q2.y, q2.z), dl));
                const f3 grdu = mk3(dot(mk3(q0.x, q0.y, q0.z), ray.d), dot(mk3(q1.x, q1.y, q1.z), ray.d), dot(mk3(q2.x, q2.y, q2.z), ray.d));
                const float l2 = dot(grdu, grdu);
                const f3 gc = cross(grdu, gro);
                const float cc = dot(gc, gc);
                if (cc < rec[4].y * l2) {
                    const float il2 = __builtin_amdgcn_rcpf(l2);
                    const float resp = response_rt(P.degree, cc * il2);
                    const float alpha = fminf(P.max_alpha, resp * q3.w);
                    const float4 q5 = rec[5], q6 = rec[6], q7 = rec[7];
                    const float hitT = oracle_order_hit_t(ray, mk3(q0.w, q1.w, q2.w), mk3(q3.x, q3.y, q3.z), mk3(q5.x, q5.y, q5.z), mk3(q6.x, q6.y, q6.z),
                                                          mk3(q7.x, q7.y, q7.z), mk3(q5.w, q6.w, q7.w));
                    if ((hitT > ray.tmin) && (hitT < ray.tmax)) {
                        if (kb.num == K) {
                            pop = true; pop_t = kb.hitT[0]; pop_a = kb.alpha[0]; pop_i = kb.idx[0];
                            kb.hitT[0] = -1.f;
                        } else {
                            kb.num++;
                        }
                        kb.insert(hitT, alpha, idx);
                    }
                }
            }
            if (BWD) {
                if (__any(pop)) {   // wave-level: the popped hits' gradients leave hit-major, lanes that popped the same particle together
                    float gd[11], wq[4], gbase[kNhtMaxIpd];
                    const bool have = pop && nht_k_pop_bwd<FAST>(P, ray, tet, density12, features, pop_t, pop_a, pop_i, bs, alive, gd, wq, gbase);
                    nht_k_bwd_flush<FAST>(P, have, pop_i, gd, wq, gbase, lane, s_kterms, g_density12, g_features);
                }
            } else if (pop) {
                nht_k_pop_fwd<FAST>(P, ray, tet, density12, features, pop_t, pop_a, pop_i, fs, alive);
            }
        }
        __syncthreads();
    }
    // drain what is left, nearest first (:343-351), as gut_render_k_body
#pragma unroll 1
    for (int step = 0; step < K; ++step) {
        const float t0 = kb.hitT[0], a0 = kb.alpha[0];
        const uint32_t i0 = kb.idx[0];
#pragma unroll
        for (int i = 0; i + 1 < K; ++i) { kb.hitT[i] = kb.hitT[i + 1]; kb.alpha[i] = kb.alpha[i + 1]; kb.idx[i] = kb.idx[i + 1]; }
        kb.hitT[K - 1] = -1.f;
        const bool act = alive && (t0 >= 0.f);
        if (BWD) {
            if (__any(act)) {
                float gd[11], wq[4], gbase[kNhtMaxIpd];
                const bool have = act && nht_k_pop_bwd<FAST>(P, ray, tet, density12, features, t0, a0, i0, bs, alive, gd, wq, gbase);
                nht_k_bwd_flush<FAST>(P, have, i0, gd, wq, gbase, lane, s_kterms, g_density12, g_features);
            }
        } else if (act) {
            nht_k_pop_fwd<FAST>(P, ray, tet, density12, features, t0, a0, i0, fs, alive);
        }
    }
    if (!BWD && ray.inside) {
        const size_t opix = (size_t)py * P.W + px;
        const size_t stride = (size_t)nr + 1;
#pragma unroll
        for (int i = 0; i < (FAST ? 24 : kNhtMaxRay); ++i) {
            if (i < nr) {
                const float v = ray.valid ? fs.acc[i] : 0.f;
                if (P.out_half) reinterpret_cast<__half*>(fd)[opix * stride + i] = __float2half(v);
                else fd[opix * stride + i] = v;
            }
        }
        const float op = ray.valid ? 1.f - fs.T : 0.f;
        if (P.out_half) reinterpret_cast<__half*>(fd)[opix * stride + nr] = __float2half(op);
        else fd[opix * stride + nr] = op;
        dist[opix] = ray.valid ? fs.D : 1e6f;
        if (P.hitcounts) out_cnt[opix] = ray.valid ? fs.cnt : 0.f;
    }
}

#include "gut_render_nht.inl"

#endif   // GRUT_RENDER_PART != 1
static uint32_t strip_grid(const GutParams& P) {
    const uint32_t tiles = (uint32_t)(P.gx * P.gy);
    return ((tiles + 7u) & ~7u) * 4u;
}
#define GRUT_DISPATCH_K(KK, ...)                               \
    switch (KK) {                                              \
    case 4: { constexpr int K_ = 4; __VA_ARGS__; } break;      \
    case 8: { constexpr int K_ = 8; __VA_ARGS__; } break;      \
    default: { constexpr int K_ = 16; __VA_ARGS__; } break;    \
    }
#if GRUT_RENDER_PART != 1
bool nht_fast_path(const GutParams& P) {
    const bool generic = getenv("GRUT_NHT_GENERIC") != nullptr;   // (development / test switch, read per call: the strip kernels for every shape)
    return P.nht && !generic && P.nht_k == kNhtK && P.nht_ipd == kNhtIpd && P.nht_support == 1 && P.nht_act == 2 && P.nht_nf == 1 && P.k_buffer == 0;
}
uint64_t nht_checkpoint_bytes(uint32_t num_boundaries) {   // (num_boundaries counts kGutSegment boundaries; the feature sweeps use every kNhtSegment-th entry)
    return ((uint64_t)nht_boundaries(num_boundaries) + 1u) * 2u * kNhtCkQuads * 64u * sizeof(float4);
}
void launch_render_nhtp_fwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const uint32_t* pos_particle,
                            const float* density12, const float* features, const float* ray_o, const float* ray_d, float* out_fd, float* out_dist,
                            float* out_cnt, void* ck_nht, const GutCheckpoints& ck, bool write_checkpoints) {
    const EntryLists lists = entry_lists(P, sorted_pos, pos_particle);
    if (write_checkpoints && ck_nht) {
        GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_nhtp_fwd_kernel<D_, true>), dim3(half_grid(P)), dim3(64), 0, s, P,
                                                          reinterpret_cast<const uint2*>(ranges), lists, reinterpret_cast<const float4*>(density12),
                                                          features, ray_o, ray_d, out_fd, out_dist, out_cnt, reinterpret_cast<float4*>(ck_nht), ck));
    } else {
        GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_nhtp_fwd_kernel<D_, false>), dim3(half_grid(P)), dim3(64), 0, s, P,
                                                          reinterpret_cast<const uint2*>(ranges), lists, reinterpret_cast<const float4*>(density12),
                                                          features, ray_o, ray_d, out_fd, out_dist, out_cnt, reinterpret_cast<float4*>(ck_nht), ck));
    }
}
void launch_render_nhtp_bwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const float* density12,
                            const float* features, const float* ray_o, const float* ray_d, const float* fd, const float* g_fd, const float* g_feat,
                            const float* g_opa, const float* dist, const float* g_dist, const GutGradSlots& slots, float* g_features,
                            const void* ck_nht, const GutCheckpoints& ck) {
    const dim3 grid(segment_grid(P, nht_boundaries(ck.num_boundaries)));
    const EntryLists lists = entry_lists(P, sorted_pos, slots.pos_particle);
    const NhtGradIn g_in{g_fd, g_feat, g_opa};
    if (g_dist) {
        GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_nhtp_bwd_kernel<D_, true>), grid, dim3(64), 0, s, P,
                                                          reinterpret_cast<const uint2*>(ranges), lists, reinterpret_cast<const float4*>(density12),
                                                          features, ray_o, ray_d, fd, g_in, dist, g_dist, slots, g_features,
                                                          reinterpret_cast<const float4*>(ck_nht), ck));
    } else {
        GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_nhtp_bwd_kernel<D_, false>), grid, dim3(64), 0, s, P,
                                                          reinterpret_cast<const uint2*>(ranges), lists, reinterpret_cast<const float4*>(density12),
                                                          features, ray_o, ray_d, fd, g_in, dist, g_dist, slots, g_features,
                                                          reinterpret_cast<const float4*>(ck_nht), ck));
    }
}
void launch_render_nht_fwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const uint32_t* pos_particle,
                           const float* density12, const float* features, const float* ray_o, const float* ray_d, float* out_fd, float* out_dist,
                           float* out_cnt) {
    const EntryLists lists = entry_lists(P, sorted_pos, pos_particle);
    if (P.k_buffer > 0) {   // the sorted hit buffer in front of the feature integration (round 6)
        const bool fast = nht_row48(P) && P.nht_act == 2 && P.nht_nf == 1;   // the default feature model
        if (fast) {
            GRUT_DISPATCH_K(P.k_buffer, hipLaunchKernelGGL((gut_render_nht_k_kernel<K_, false, true>), dim3(strip_grid(P)), dim3(64), 0, s, P,
                                                           reinterpret_cast<const uint2*>(ranges), lists, reinterpret_cast<const float4*>(density12), features,
                                                           ray_o, ray_d, out_fd, out_dist, out_cnt, nullptr, nullptr, nullptr, nullptr));
        } else {
            GRUT_DISPATCH_K(P.k_buffer, hipLaunchKernelGGL((gut_render_nht_k_kernel<K_, false, false>), dim3(strip_grid(P)), dim3(64), 0, s, P,
                                                           reinterpret_cast<const uint2*>(ranges), lists, reinterpret_cast<const float4*>(density12), features,
                                                           ray_o, ray_d, out_fd, out_dist, out_cnt, nullptr, nullptr, nullptr, nullptr));
        }
        return;
    }
    hipLaunchKernelGGL(gut_render_nht_fwd_kernel, dim3(strip_grid(P)), dim3(64), 0, s, P, reinterpret_cast<const uint2*>(ranges), lists,
                       reinterpret_cast<const float4*>(density12), features, ray_o, ray_d, out_fd, out_dist, out_cnt);
}
void launch_render_nht_bwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const uint32_t* pos_particle,
                           const float* density12, const float* features, const float* ray_o, const float* ray_d, const float* fd, const float* g_fd,
                           const float* dist, const float* g_dist, float* g_density12, float* g_features) {
    const EntryLists lists = entry_lists(P, sorted_pos, pos_particle);
    if (P.k_buffer > 0) {
        const bool fast = nht_row48(P) && P.nht_act == 2 && P.nht_nf == 1;
        if (fast) {
            GRUT_DISPATCH_K(P.k_buffer, hipLaunchKernelGGL((gut_render_nht_k_kernel<K_, true, true>), dim3(strip_grid(P)), dim3(64), 0, s, P,
                                                           reinterpret_cast<const uint2*>(ranges), lists, reinterpret_cast<const float4*>(density12), features,
                                                           ray_o, ray_d, const_cast<float*>(fd), const_cast<float*>(dist), nullptr, g_fd, g_dist, g_density12,
                                                           g_features));
        } else {
            GRUT_DISPATCH_K(P.k_buffer, hipLaunchKernelGGL((gut_render_nht_k_kernel<K_, true, false>), dim3(strip_grid(P)), dim3(64), 0, s, P,
                                                           reinterpret_cast<const uint2*>(ranges), lists, reinterpret_cast<const float4*>(density12), features,
                                                           ray_o, ray_d, const_cast<float*>(fd), const_cast<float*>(dist), nullptr, g_fd, g_dist, g_density12,
                                                           g_features));
        }
        return;
    }
    hipLaunchKernelGGL(gut_render_nht_bwd_kernel, dim3(strip_grid(P)), dim3(64), 0, s, P, reinterpret_cast<const uint2*>(ranges), lists,
                       reinterpret_cast<const float4*>(density12), features, ray_o, ray_d, fd, g_fd, dist, g_dist, g_density12, g_features);
}
#endif   // GRUT_RENDER_PART != 1
#if GRUT_RENDER_PART != 0   // the sorted hit buffer's launchers (and, through them, its kernels' instantiations): see the note at the top
void launch_render_k_fwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const uint32_t* pos_particle,
                         const float* density12, const float* rgb, const float* ray_o, const float* ray_d, float* out_fd, float* out_dist,
                         float* out_cnt) {
    const EntryLists lists = entry_lists(P, sorted_pos, pos_particle);
    GRUT_DISPATCH_K(P.k_buffer, hipLaunchKernelGGL((gut_render_k_fwd_kernel<K_>), dim3(strip_grid(P)), dim3(64), 0, s, P,
                                                   reinterpret_cast<const uint2*>(ranges), lists, reinterpret_cast<const float4*>(density12), rgb,
                                                   ray_o, ray_d, reinterpret_cast<float4*>(out_fd), out_dist, out_cnt));
}
void launch_render_k_bwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const uint32_t* pos_particle,
                         const float* density12, const float* rgb, const float* ray_o, const float* ray_d, const float* fd, const float* g_fd,
                         const float* dist, const float* g_dist, float* g_density12, float* g_rgb) {
    const EntryLists lists = entry_lists(P, sorted_pos, pos_particle);
    GRUT_DISPATCH_K(P.k_buffer, hipLaunchKernelGGL((gut_render_k_bwd_kernel<K_>), dim3(strip_grid(P)), dim3(64), 0, s, P,
                                                   reinterpret_cast<const uint2*>(ranges), lists, reinterpret_cast<const float4*>(density12), rgb,
                                                   ray_o, ray_d, reinterpret_cast<float4*>(const_cast<float*>(fd)), const_cast<float*>(dist),
                                                   reinterpret_cast<const float4*>(g_fd), g_dist, g_density12, g_rgb));
}

#endif   // GRUT_RENDER_PART != 0
}  // namespace grut
