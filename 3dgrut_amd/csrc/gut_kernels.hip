// gut_kernels.hip — 3DGUT device code for gfx950: unscented projection onto tiles, ordered tile
// expansion, tile ranges, gradient gather and projection backward (the compositing sweeps: gut_render.hip).
//
// Reference behaviour restated (not translated): threedgut_tracer/include/3dgut/kernels/cuda/renderers/
// gutProjector.cuh:32-430, gutKBufferRenderer.cuh:199-352,642-716, common/rayPayload*.cuh,
// models/gaussianParticles.cuh:484-751.  CDNA4 design:
//   * binning key is (tile, depth-rank): particles are depth-sorted once (N keys), tile entries are
//     emitted in rank order, so only the tile bits need stable radix passes over the I entries;
//   * the per-particle tile walks (culling count, expansion) are wave-cooperative: boxes of up to 4x4 tiles
//     by one 16-lane row each, up to 8x4 by half a wave, larger ones by the whole wave;
//   * the sweeps write one gradient slot per (tile entry, half tile); the gather kernel sums a
//     particle's slots with four lanes per particle and contracts them into the parameter gradients.
#include <cstdlib>

#include <hip/hip_fp16.h>

#include "gut_internal.hpp"

namespace grut {

namespace {

// ---------------------------------------------------------------------------------------------
// SH (models/gaussianParticles.cuh:61-100)
// ---------------------------------------------------------------------------------------------
#define GRUT_SH_CONSTANTS                                                                                                   \
    constexpr float kC0 = 0.28209479177387814f, kC1 = 0.4886025119029199f;                                                  \
    constexpr float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,        \
                              0.5462742152960396f};                                                                         \
    constexpr float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,          \
                              -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};                               \
    (void)kC0; (void)kC1; (void)kC2; (void)kC3

__device__ __forceinline__ void sh_basis(int deg, f3 d, float b[16]) {
    GRUT_SH_CONSTANTS;
    const float x = d.x, y = d.y, z = d.z;
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = 0.f;
    b[0] = kC0;
    if (deg > 0) {
        b[1] = -kC1 * y; b[2] = kC1 * z; b[3] = -kC1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = kC2[0] * xy; b[5] = kC2[1] * yz; b[6] = kC2[2] * (2.f * zz - xx - yy); b[7] = kC2[3] * xz; b[8] = kC2[4] * (xx - yy);
            if (deg > 2) {
                b[9]  = kC3[0] * y * (3.f * xx - yy);
                b[10] = kC3[1] * xy * z;
                b[11] = kC3[2] * y * (4.f * zz - xx - yy);
                b[12] = kC3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = kC3[4] * x * (4.f * zz - xx - yy);
                b[14] = kC3[5] * z * (xx - yy);
                b[15] = kC3[6] * x * (xx - 3.f * yy);
            }
        }
    }
}
// d basis / d direction
__device__ __forceinline__ void sh_basis_grad(int deg, f3 d, f3 g[16]) {
    GRUT_SH_CONSTANTS;
    const float x = d.x, y = d.y, z = d.z;
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = mk3(0.f, 0.f, 0.f);
    if (deg > 0) {
        g[1] = mk3(0.f, -kC1, 0.f); g[2] = mk3(0.f, 0.f, kC1); g[3] = mk3(-kC1, 0.f, 0.f);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            g[4] = kC2[0] * mk3(y, x, 0.f);
            g[5] = kC2[1] * mk3(0.f, z, y);
            g[6] = kC2[2] * mk3(-2.f * x, -2.f * y, 4.f * z);
            g[7] = kC2[3] * mk3(z, 0.f, x);
            g[8] = kC2[4] * mk3(2.f * x, -2.f * y, 0.f);
            if (deg > 2) {
                g[9]  = kC3[0] * mk3(6.f * xy, 3.f * xx - 3.f * yy, 0.f);
                g[10] = kC3[1] * mk3(yz, xz, xy);
                g[11] = kC3[2] * mk3(-2.f * xy, 4.f * zz - xx - 3.f * yy, 8.f * yz);
                g[12] = kC3[3] * mk3(-6.f * xz, -6.f * yz, 6.f * zz - 3.f * xx - 3.f * yy);
                g[13] = kC3[4] * mk3(4.f * zz - 3.f * xx - yy, -2.f * xy, 8.f * xz);
                g[14] = kC3[5] * mk3(2.f * xz, -2.f * yz, xx - yy);
                g[15] = kC3[6] * mk3(3.f * xx - 3.f * yy, -6.f * xy, 0.f);
            }
        }
    }
}

// The backward's copies of the two functions above with every rounding pinned in the source (opaque products, explicit fused multiply-adds).
// Three kernels evaluate the basis for the gradient - the fused gather (a quad's lanes keep four functions each), the projection backward
// (all sixteen) and the rebuild from gathered view factors - and the data-parallel contract is that they agree BIT FOR BIT (a replica that
// rebuilds the SH gradient from factors must hold what a single GPU holds).  Left to the compiler, a product such as 3 x^2 is contracted into
// the neighbouring subtraction when it has one use and kept when it has several, i.e. differently in a kernel that needs four functions
// than in one that needs sixteen (round 5: tests/test_gut_gpu.py::test_factored_backward_rebuilds_the_sph_gradient[3-2]).
__device__ __forceinline__ void sh_basis_pinned(int deg, f3 d, float b[16]) {
    GRUT_SH_CONSTANTS;
    const float x = d.x, y = d.y, z = d.z;
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = 0.f;
    b[0] = kC0;
    if (deg > 0) {
        b[1] = -kC1 * y; b[2] = kC1 * z; b[3] = -kC1 * x;
        if (deg > 1) {
            const float xx = mul_rn(x, x), yy = mul_rn(y, y), zz = mul_rn(z, z), xy = mul_rn(x, y), yz = mul_rn(y, z), xz = mul_rn(x, z);
            const float t6 = add_rn(fmaf(2.f, zz, -xx), -yy), d8 = add_rn(xx, -yy);
            b[4] = kC2[0] * xy; b[5] = kC2[1] * yz; b[6] = kC2[2] * t6; b[7] = kC2[3] * xz; b[8] = kC2[4] * d8;
            if (deg > 2) {
                const float t11 = add_rn(fmaf(4.f, zz, -xx), -yy);
                b[9]  = kC3[0] * y * fmaf(3.f, xx, -yy);
                b[10] = kC3[1] * xy * z;
                b[11] = kC3[2] * y * t11;
                b[12] = kC3[3] * z * fmaf(-3.f, yy, fmaf(-3.f, xx, mul_rn(2.f, zz)));
                b[13] = kC3[4] * x * t11;
                b[14] = kC3[5] * z * d8;
                b[15] = kC3[6] * x * fmaf(-3.f, yy, xx);
            }
        }
    }
}
__device__ __forceinline__ void sh_basis_grad_pinned(int deg, f3 d, f3 g[16]) {
    GRUT_SH_CONSTANTS;
    const float x = d.x, y = d.y, z = d.z;
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = mk3(0.f, 0.f, 0.f);
    if (deg > 0) {
        g[1] = mk3(0.f, -kC1, 0.f); g[2] = mk3(0.f, 0.f, kC1); g[3] = mk3(-kC1, 0.f, 0.f);
        if (deg > 1) {
            const float xx = mul_rn(x, x), yy = mul_rn(y, y), zz = mul_rn(z, z), xy = mul_rn(x, y), yz = mul_rn(y, z), xz = mul_rn(x, z);
            g[4] = kC2[0] * mk3(y, x, 0.f);
            g[5] = kC2[1] * mk3(0.f, z, y);
            g[6] = kC2[2] * mk3(-2.f * x, -2.f * y, 4.f * z);
            g[7] = kC2[3] * mk3(z, 0.f, x);
            g[8] = kC2[4] * mk3(2.f * x, -2.f * y, 0.f);
            if (deg > 2) {
                const float t9 = fmaf(3.f, xx, -mul_rn(3.f, yy));
                g[9]  = kC3[0] * mk3(6.f * xy, t9, 0.f);
                g[10] = kC3[1] * mk3(yz, xz, xy);
                g[11] = kC3[2] * mk3(-2.f * xy, fmaf(-3.f, yy, fmaf(4.f, zz, -xx)), 8.f * yz);
                g[12] = kC3[3] * mk3(-6.f * xz, -6.f * yz, fmaf(-3.f, yy, fmaf(-3.f, xx, mul_rn(6.f, zz))));
                g[13] = kC3[4] * mk3(add_rn(fmaf(-3.f, xx, mul_rn(4.f, zz)), -yy), -2.f * xy, 8.f * xz);
                g[14] = kC3[5] * mk3(2.f * xz, -2.f * yz, add_rn(xx, -yy));
                g[15] = kC3[6] * mk3(t9, -6.f * xy, 0.f);
            }
        }
    }
}
// the rest of the projection backward's chain, shared by the same kernels for the same reason
__device__ __forceinline__ f3 sh_view_direction(f3 p, f3 cam, float& ilen) {
    const f3 v = mk3(add_rn(p.x, -cam.x), add_rn(p.y, -cam.y), add_rn(p.z, -cam.z));
    ilen = 1.f / sqrtf(fmaf(v.z, v.z, fmaf(v.y, v.y, mul_rn(v.x, v.x))));
    return mk3(mul_rn(v.x, ilen), mul_rn(v.y, ilen), mul_rn(v.z, ilen));
}
__device__ __forceinline__ float sh_row_dot(f3 g, float m0, float m1, float m2) { return fmaf(g.z, m2, fmaf(g.y, m1, mul_rn(g.x, m0))); }
__device__ __forceinline__ f3 sh_axpy(f3 acc, f3 db, float s) { return mk3(fmaf(db.x, s, acc.x), fmaf(db.y, s, acc.y), fmaf(db.z, s, acc.z)); }
__device__ __forceinline__ f3 sh_sum(f3 a, f3 b) { return mk3(add_rn(a.x, b.x), add_rn(a.y, b.y), add_rn(a.z, b.z)); }
__device__ __forceinline__ f3 sh_position_gradient(f3 dir, float ilen, f3 gdir) {   // safe_normalize backward
    const float ng = fmaf(dir.z, gdir.z, fmaf(dir.y, gdir.y, mul_rn(dir.x, gdir.x)));
    return mk3(mul_rn(fmaf(-dir.x, ng, gdir.x), ilen), mul_rn(fmaf(-dir.y, ng, gdir.y), ilen), mul_rn(fmaf(-dir.z, ng, gdir.z), ilen));
}

// ---------------------------------------------------------------------------------------------
// tile-space helpers (gutProjector.cuh:32-116)
// ---------------------------------------------------------------------------------------------
struct TileBBox {
    int minx, miny, maxx, maxy;
};
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ TileBBox tile_space_bbox(int gx, int gy, float px, float py, float ex, float ey) {
    constexpr float inv = 1.f / 16.f;  // exact power of two
    TileBBox b;
    b.minx = clampi((int)floorf((px - 0.5f - ex) * inv), 0, gx);
    b.miny = clampi((int)floorf((py - 0.5f - ey) * inv), 0, gy);
    b.maxx = clampi((int)ceilf((px - 0.5f + ex) * inv), 0, gx);
    b.maxy = clampi((int)ceilf((py - 0.5f + ey) * inv), 0, gy);
    return b;
}
__device__ __forceinline__ float saturate(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

// tileMinParticlePowerResponse, gutProjector.cuh:49-78.  Evaluated by both the counting pass and the
// expansion pass on identical stored inputs.  FP contraction is switched off inside so that both kernels
// round identically whatever the surrounding code looks like after inlining (a disagreement would only
// cost a padded or a dropped tile entry: the expansion bounds its writes by the counted range).
struct TileConic {   // per-particle constants of the test; rcpx / rcpy are the two divisions of :63-64, done once per particle
    float cx, cy, cz, rcpx, rcpy;
};
__device__ __forceinline__ TileConic tile_conic(float4 co) {
#pragma clang fp contract(off)
    const float ts = 16.f;
    return {co.x, co.y, co.z, 1.f / (ts * ts * co.x), 1.f / (ts * ts * co.z)};
}
__device__ __forceinline__ float tile_min_power(float tx, float ty, const TileConic& co, float mx, float my) {
#pragma clang fp contract(off)
    const float ts = 16.f;
    const float tminx = ts * tx, tminy = ts * ty, tmaxx = ts + tminx, tmaxy = ts + tminy;
    const float offx = tminx - mx, offy = tminy - my;
    const float lax = offx > 0.f ? 1.f : 0.f, lay = offy > 0.f ? 1.f : 0.f;
    const float nrx = lax + (mx > tmaxx ? 1.f : 0.f), nry = lay + (my > tmaxy ? 1.f : 0.f);
    if ((nrx + nry) > 0.f) {
        const float px = lax > 0.f ? tminx : tmaxx, py = lay > 0.f ? tminy : tmaxy;
        const float dx = copysignf(ts, offx), dy = copysignf(ts, offy);
        const float diffx = mx - px, diffy = my - py;
        const float tx_ = nry * saturate((dx * co.cx * diffx + dx * co.cy * diffy) * co.rcpx);
        const float ty_ = nrx * saturate((dy * co.cy * diffx + dy * co.cz * diffy) * co.rcpy);
        const float mdx = mx - (px + tx_ * dx), mdy = my - (py + ty_ * dy);
        return 0.5f * (co.cx * mdx * mdx + co.cz * mdy * mdy) + co.cy * mdx * mdy;
    }
    return 0.f;
}

// Wave-cooperative walk over one particle's tile bounding box: the 64 lanes form an 8x8 block of tiles that
// sweeps the box, so a particle covering thousands of tiles costs the wave ~area/64 steps instead of stalling one
// lane for `area` steps (projected extents are heavy-tailed: kernel time used to be one giant particle's lane).
// All arguments are wave-uniform.  `emit(keep, tile_index)` is called once per step by every lane.
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float bcast_f(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
template <typename Emit>
__device__ __forceinline__ void coop_tile_walk(int lane, int gx, bool culling, TileBBox bb, const TileConic& co, float cx, float cy, float pmax,
                                               Emit&& emit) {
    const int lx = lane & 7, ly = lane >> 3;
    for (int by = bb.miny; by < bb.maxy; by += 8)
        for (int bx = bb.minx; bx < bb.maxx; bx += 8) {
            const int x = bx + lx, y = by + ly;
            bool keep = (x < bb.maxx) && (y < bb.maxy);
            if (keep && culling) keep = tile_min_power((float)x, (float)y, co, cx, cy) < pmax;
            emit(keep, (uint32_t)(y * gx + x));
        }
}

// Most boxes are a few tiles wide, where an 8x8 block leaves most lanes idle: boxes of at most kRowWalkSide x kRowWalkSide tiles
// are walked by ONE 16-lane row as a 4x4 block, the four rows of the wave working on four different particles at once.
// The arguments are uniform per row (the row's current particle); `active` rows take part, all lanes call `emit(keep,
// tile)` once per step (it may ballot).
constexpr int kRowWalkSide = 4;   // boxes up to 4x4 tiles: one step of a 16-lane row
__device__ __forceinline__ int row_bcast_i(int v, int src_lane) { return __shfl(v, src_lane, 64); }
__device__ __forceinline__ float row_bcast_f(float v, int src_lane) { return __shfl(v, src_lane, 64); }
template <typename Emit>
__device__ __forceinline__ void row_tile_walk(int lane, int gx, bool culling, bool active, TileBBox bb, const TileConic& co, float cx, float cy,
                                              float pmax, Emit&& emit) {
    const int lx = lane & 3, ly = (lane >> 2) & 3;
    int bx = bb.minx, by = bb.miny;
    bool run = active && bb.minx < bb.maxx && bb.miny < bb.maxy;
    while (__ballot(run)) {
        const int x = bx + lx, y = by + ly;
        bool keep = run && (x < bb.maxx) && (y < bb.maxy);
        if (keep && culling) keep = tile_min_power((float)x, (float)y, co, cx, cy) < pmax;
        emit(keep, (uint32_t)(y * gx + x));
        bx += 4;
        if (bx >= bb.maxx) {
            bx = bb.minx;
            by += 4;
            run = run && by < bb.maxy;
        }
    }
}
__device__ __forceinline__ int bbox_area(const TileBBox& b) { return (b.maxx - b.minx) * (b.maxy - b.miny); }
__device__ __forceinline__ bool bbox_fits_row(const TileBBox& b) { return (b.maxx - b.minx) <= kRowWalkSide && (b.maxy - b.miny) <= kRowWalkSide; }
// Boxes of at most 8x4 or 4x8 tiles (27 % of the visible particles of the bench cloud, next to 46 % that fit a row) take one
// step of HALF a wave, laid out as an 8x4 or a 4x8 block: the two halves work on two different particles at once.
__device__ __forceinline__ bool bbox_fits_half(const TileBBox& b) {
    const int w = b.maxx - b.minx, h = b.maxy - b.miny;
    return (w <= 8 && h <= 4) || (w <= 4 && h <= 8);
}
// tile of this lane inside a half-wave block over box b (arguments uniform per half); false when the lane falls outside
__device__ __forceinline__ bool half_block_tile(int lane, const TileBBox& b, int& x, int& y) {
    const int hl = lane & 31;
    const bool wide = (b.maxx - b.minx) > 4;
    x = b.minx + (wide ? (hl & 7) : (hl & 3));
    y = b.miny + (wide ? (hl >> 3) : (hl >> 2));
    return (x < b.maxx) && (y < b.maxy);
}
// The j-th member of a class inside each group of lanes (16-lane rows, 32-lane halves): `member` lanes are ranked inside their
// group, and step j of a walk serves the rank-j member of every group at once — max-over-groups steps instead of one step
// per lane position.  Returns the source lane of the caller's group for step j (valid when `act`).
struct GroupPick {
    int rank;        // rank of this lane among the members of its group (meaningful for members)
    int steps;       // max over groups of the member count
};
template <int GROUP>
__device__ __forceinline__ GroupPick group_rank(int lane, bool member) {
    const unsigned long long m = __ballot(member);
    const int shift = lane & ~(GROUP - 1) & 63;
    const unsigned long long gmask = (GROUP == 64) ? ~0ull : ((1ull << GROUP) - 1ull);
    const unsigned long long mine = (m >> shift) & gmask;
    GroupPick g;
    g.rank = __popcll(mine & ((1ull << (lane & (GROUP - 1))) - 1ull));
    int steps = 0;
#pragma unroll
    for (int s = 0; s < 64; s += GROUP) steps = max(steps, (int)__popcll((m >> s) & gmask));
    g.steps = steps;
    return g;
}
template <int GROUP>
__device__ __forceinline__ int group_source(int lane, bool member, const GroupPick& g, int j, bool& act) {
    const unsigned long long sel = __ballot(member && g.rank == j);   // at most one lane per group
    const int shift = lane & ~(GROUP - 1) & 63;
    const unsigned long long gmask = (GROUP == 64) ? ~0ull : ((1ull << GROUP) - 1ull);
    const unsigned long long mine = (sel >> shift) & gmask;
    act = mine != 0;
    return shift | (act ? (__ffsll((long long)mine) - 1) : 0);
}

// ---------------------------------------------------------------------------------------------
// K1: projection onto tiles — GUTProjector::eval (gutProjector.cuh:217-322)
// ---------------------------------------------------------------------------------------------
// 6 waves/SIMD measured best (0.227 ms; 4: 0.239, 5: 0.232, 8: 0.261 with spills).  One instantiation per (camera model, global /
// rolling shutter): with the three models and the shutter iterations in one body the kernel was 17 k instructions (140 KB of code
// against a 64 KB instruction cache shared by two CUs).
template <int MODEL, int ROLLING>
#ifndef GRUT_PROJECT_WAVES_MAX
#define GRUT_PROJECT_WAVES_MAX 6   // (8 fits the specialised kernels, 60 VGPRs, and changes nothing: 0.122 vs 0.120 ms)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, GRUT_PROJECT_WAVES_MAX))) void gut_project_kernel(GutParams P, const float4* __restrict__ density12,
                                                          const float* __restrict__ sph, GutProjected out,
                                                          int32_t* __restrict__ visibility, uint32_t* __restrict__ num_visible) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const FramePoses& FP = frame_poses(P);
    bool has_tiles = false;
    uint32_t ntiles = 0;
    int vis = 0;
    float cx = 0.f, cy = 0.f, ex = 0.f, ey = 0.f, depth = 0.f, pmax_tile = 0.f, view_z_keep = 0.f;
    float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
    TileBBox bb = {0, 0, 0, 0};
    f3 pos = mk3(0.f, 0.f, 0.f);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
    if (i < P.N) {
        a = density12[3 * (size_t)i + 0];  // pos.xyz, density
        b = density12[3 * (size_t)i + 1];  // quat wxyz
        c = density12[3 * (size_t)i + 2];  // scale.xyz, pad
        pos = mk3(a.x, a.y, a.z);
        const float opacity = a.w;

        // The depth is the sort key: its BITS define the per-tile compositing order, so it is evaluated with separately rounded
        // multiplies and adds in the reference's source order, dot(R.row2, pos) + t.z (gutProjector.cuh:131-140, 315-321) — never
        // contracted into FMAs, whatever -ffp-contract says — and the CPU checker reproduces every key bit for bit.
        const float view_z = add_rn_u(FP.view_t[2], add_rn(add_rn(mul_rn_u(FP.view_R[6], pos.x), mul_rn_u(FP.view_R[7], pos.y)), mul_rn_u(FP.view_R[8], pos.z)));
        view_z_keep = view_z;
        bool ok = (opacity >= P.min_alpha) && (view_z >= 0.2f);
        if (ok) {
            const m3 rotT = quat_wxyz_to_rotT(b.x, b.y, b.z, b.w);
            float spx[7], spy[7];
            int nvalid = 0;
            nvalid += project_point_with_shutter<MODEL, ROLLING>(P.cam, FP, P.n_rs_iter, pos, P.ut_margin, spx[0], spy[0]) ? 1 : 0;
            cx = spx[0] * P.ut_w0m; cy = spy[0] * P.ut_w0m;
            const f3 axes[3] = {rotT.r0 * (P.ut_delta * c.x), rotT.r1 * (P.ut_delta * c.y), rotT.r2 * (P.ut_delta * c.z)};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                nvalid += project_point_with_shutter<MODEL, ROLLING>(P.cam, FP, P.n_rs_iter, pos + axes[k], P.ut_margin, spx[k + 1], spy[k + 1]) ? 1 : 0;
                cx += P.ut_wi * spx[k + 1]; cy += P.ut_wi * spy[k + 1];
                nvalid += project_point_with_shutter<MODEL, ROLLING>(P.cam, FP, P.n_rs_iter, pos - axes[k], P.ut_margin, spx[k + 4], spy[k + 4]) ? 1 : 0;
                cx += P.ut_wi * spx[k + 4]; cy += P.ut_wi * spy[k + 4];
            }
            ok = P.ut_require_all ? (nvalid == 7) : (nvalid > 0);
            if (ok) {
                float cov0, cov1, cov2;
                {
                    const float dx = spx[0] - cx, dy = spy[0] - cy;
                    cov0 = P.ut_w0c * dx * dx; cov1 = P.ut_w0c * dx * dy; cov2 = P.ut_w0c * dy * dy;
                }
#pragma unroll
                for (int k = 1; k < 7; ++k) {
                    const float dx = spx[k] - cx, dy = spy[k] - cy;
                    cov0 += P.ut_wi * dx * dx; cov1 += P.ut_wi * dx * dy; cov2 += P.ut_wi * dy * dy;
                }
                // computeProjectedExtentConicOpacity (:81-116)
                const float ca = cov0 + 0.3f, cb = cov1, cc = cov2 + 0.3f;
                const float det = ca * cc - cb * cb;
                ok = det != 0.f;
                if (ok) {
                    const float idet = 1.f / det;
                    co.x = cc * idet; co.y = -cb * idet; co.z = ca * idet;
                    const float cov_det = cov0 * cov2 - cov1 * cov1;
                    co.w = opacity * sqrtf(fmaxf(0.000025f, cov_det * idet));
                    ok = co.w >= P.min_alpha;
                    if (ok) {
                        const float pmax = logf(co.w / P.min_alpha);
                        const float ef = P.tight_opacity ? fminf(3.33f, sqrtf(2.f * pmax)) : 3.33f;
                        const float mid = 0.5f * (ca + cc);
                        const float lambda = mid + sqrtf(fmaxf(0.01f, mid * mid - det));
                        const float radius = ef * sqrtf(lambda);
                        ex = P.rect_bounding ? fminf(ef * sqrtf(ca), radius) : radius;
                        ey = P.rect_bounding ? fminf(ef * sqrtf(cc), radius) : radius;
                        ok = radius > 0.f;
                        if (ok) {
                            vis = 1;
                            bb = tile_space_bbox(P.gx, P.gy, cx, cy, ex, ey);
                            pmax_tile = logf(co.w / P.min_alpha);  // same expression as the expansion pass
                            if (!P.tile_culling) ntiles = (uint32_t)((bb.maxx - bb.minx) * (bb.maxy - bb.miny));
                        }
                    }
                }
            }
        }
    }
    // per-tile culling count (gutProjector.cuh:279-293): small boxes four at a time (one per 16-lane row), the heavy
    // tail one particle at a time across the whole wave
    // (round 6) the walks that take ONE step - boxes up to 4x4 tiles on a 16-lane row, up to 8x4 / 4x8 on a half wave: 73 % of the bench
    // cloud's visible particles - leave their keep mask (bit = the lane's position in the row / half) and the box next to the count
    // (GutProjected::walk8): the expansion reads 8 bytes per such particle instead of 32 from three arrays and does not evaluate the
    // tile test a second time
    uint32_t walk_mask = 0u, walk_kind = 0u;
    if (P.tile_culling) {
        const TileConic tc = tile_conic(co);
        const int area = vis ? bbox_area(bb) : 0;
        const bool small = area > 0 && bbox_fits_row(bb);
        const int row_shift = lane & 48;
        const GroupPick rows = group_rank<16>(lane, small);
        for (int k = 0; k < rows.steps; ++k) {
            bool act;
            const int src = group_source<16>(lane, small, rows, k, act);
            const TileBBox sb = {row_bcast_i(bb.minx, src), row_bcast_i(bb.miny, src), row_bcast_i(bb.maxx, src), row_bcast_i(bb.maxy, src)};
            const TileConic sco = {row_bcast_f(tc.cx, src), row_bcast_f(tc.cy, src), row_bcast_f(tc.cz, src), row_bcast_f(tc.rcpx, src),
                                   row_bcast_f(tc.rcpy, src)};
            uint32_t cnt = 0, m16 = 0u;
            row_tile_walk(lane, P.gx, true, act, sb, sco, row_bcast_f(cx, src), row_bcast_f(cy, src), row_bcast_f(pmax_tile, src),
                          [&](bool keep, uint32_t) { m16 = (uint32_t)((__ballot(keep) >> row_shift) & 0xFFFFull); cnt += (uint32_t)__popc(m16); });
            if (lane == src && act) { ntiles = cnt; walk_mask = m16; walk_kind = 1u; }
        }
        const bool halfc = area > 0 && !small && bbox_fits_half(bb);
        const int half_shift = lane & 32;
        const GroupPick halves = group_rank<32>(lane, halfc);
        for (int k = 0; k < halves.steps; ++k) {
            bool act;
            const int src = group_source<32>(lane, halfc, halves, k, act);
            const TileBBox sb = {row_bcast_i(bb.minx, src), row_bcast_i(bb.miny, src), row_bcast_i(bb.maxx, src), row_bcast_i(bb.maxy, src)};
            const TileConic sco = {row_bcast_f(tc.cx, src), row_bcast_f(tc.cy, src), row_bcast_f(tc.cz, src), row_bcast_f(tc.rcpx, src),
                                   row_bcast_f(tc.rcpy, src)};
            const float scx = row_bcast_f(cx, src), scy = row_bcast_f(cy, src), spmax = row_bcast_f(pmax_tile, src);
            int x, y;
            bool keep = half_block_tile(lane, sb, x, y) && act;
            if (keep) keep = tile_min_power((float)x, (float)y, sco, scx, scy) < spmax;
            const uint32_t m32 = (uint32_t)((__ballot(keep) >> half_shift) & 0xFFFFFFFFull);
            if (lane == src && act) { ntiles = (uint32_t)__popc(m32); walk_mask = m32; walk_kind = 2u; }
        }
        unsigned long long todo = __ballot(area > 0 && !small && !halfc);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const TileBBox sb = {bcast_i(bb.minx, src), bcast_i(bb.miny, src), bcast_i(bb.maxx, src), bcast_i(bb.maxy, src)};
            const TileConic sco = {bcast_f(tc.cx, src), bcast_f(tc.cy, src), bcast_f(tc.cz, src), bcast_f(tc.rcpx, src), bcast_f(tc.rcpy, src)};
            uint32_t cnt = 0;
            coop_tile_walk(lane, P.gx, true, sb, sco, bcast_f(cx, src), bcast_f(cy, src), bcast_f(pmax_tile, src),
                           [&](bool keep, uint32_t) { cnt += (uint32_t)__popcll(__ballot(keep)); });
            if (lane == src) ntiles = cnt;
        }
    }
    if (i < P.N) {
        visibility[i] = vis;
        out.tiles_count[i] = ntiles;
        has_tiles = ntiles > 0;
        if (has_tiles && out.walk8)   // {minx : 12 | miny : 12 | width - 1 : 3 | height - 1 : 3 | kind : 2, keep mask}; kind 0 = the expansion walks the box itself
            out.walk8[i] = make_uint2((uint32_t)bb.minx | ((uint32_t)bb.miny << 12) | ((uint32_t)(bb.maxx - bb.minx - 1) & 7u) << 24 |
                                          ((uint32_t)(bb.maxy - bb.miny - 1) & 7u) << 27 | ((ex <= 1e-06f ? 0u : walk_kind) << 30), walk_mask);   // (an extent the expansion skips, gutProjector.cuh:344-348: its own path)
        if (!has_tiles) {
            out.proj_pos[i] = make_float2(0.f, 0.f);
            out.conic_opacity[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            out.extent[i] = make_float2(0.f, 0.f);
            out.depth[i] = 0.f;
            out.depth_key[i] = 0xFFFFFFFFu;  // sorts behind every visible particle
        } else {
            const f3 ray = pos - mk3(FP.s2w_t[0], FP.s2w_t[1], FP.s2w_t[2]);
            // (also a sort key, with global_z_order off: same rule)
            const float dist = __fsqrt_rn(add_rn(add_rn(mul_rn(ray.x, ray.x), mul_rn(ray.y, ray.y)), mul_rn(ray.z, ray.z)));
            const f3 dir = ray * (1.f / dist);
            float basis[16];
            sh_basis(P.n_active, dir, basis);
            const int nact = (P.n_active + 1) * (P.n_active + 1);
            const float* coef = sph + (size_t)i * 3 * P.ncoef;
            float r = 0.f, g = 0.f, bl = 0.f;
            if (P.nht) {        // per-ray features: no per-particle radiance (gutProjector.cuh:306)
                r = g = bl = -0.5f;
            } else if (P.sph_half) {   // PARTICLE_FEATURE_HALF: the coefficients are stored as IEEE half, the arithmetic stays fp32
                const __half* hc = reinterpret_cast<const __half*>(sph) + (size_t)i * 3 * P.ncoef;
                for (int k = 0; k < nact && k < P.ncoef; ++k) {
                    r = fmaf(basis[k], __half2float(hc[3 * k + 0]), r);
                    g = fmaf(basis[k], __half2float(hc[3 * k + 1]), g);
                    bl = fmaf(basis[k], __half2float(hc[3 * k + 2]), bl);
                }
            } else if (P.ncoef == 16) {
                // a lane reads its own 192-byte row: 16-byte loads (12 requests per row instead of 48), accumulated in the
                // same coefficient order as the scalar loop
                const float4* row = reinterpret_cast<const float4*>(coef);
                float acc[3] = {0.f, 0.f, 0.f};
                const int nq = (3 * nact + 3) >> 2;
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    if (q < nq) {
                        const float4 v = row[q];
                        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int k = (4 * q + j) / 3, c = (4 * q + j) % 3;
                            if (k < nact) acc[c] = fmaf(basis[k], e[j], acc[c]);
                        }
                    }
                }
                r = acc[0]; g = acc[1]; bl = acc[2];
            } else {
                for (int k = 0; k < nact && k < P.ncoef; ++k) {
                    r = fmaf(basis[k], coef[3 * k + 0], r);
                    g = fmaf(basis[k], coef[3 * k + 1], g);
                    bl = fmaf(basis[k], coef[3 * k + 2], bl);
                }
            }
            out.rgb[3 * (size_t)i + 0] = r + 0.5f;
            out.rgb[3 * (size_t)i + 1] = g + 0.5f;
            out.rgb[3 * (size_t)i + 2] = bl + 0.5f;
            if (P.rec64) {   // everything the compositing sweeps need of this particle, on one 64-byte line (scale.w: the expansion fills in part_offset)
                float4* rec = out.rec64 + 4 * (size_t)i;
                rec[0] = a; rec[1] = b; rec[2] = make_float4(c.x, c.y, c.z, 0.f);
                rec[3] = make_float4(r + 0.5f, g + 0.5f, bl + 0.5f, 0.f);
            }
            out.proj_pos[i] = make_float2(cx, cy);
            out.conic_opacity[i] = co;
            out.extent[i] = make_float2(ex, ey);
            depth = P.global_z ? view_z_keep : dist;
            out.depth[i] = depth;
            out.depth_key[i] = __float_as_uint(depth);
        }
    }   // (particle_idx, the payload of the depth sort, is an iota: generated by the sort's first pass)
    // Nv for the byte model: one atomic per wave
    const unsigned long long m = __ballot(has_tiles);
    if (lane_id() == 0 && m) atomicAdd(num_visible + (blockIdx.x % kGutCounterReplicas) * kGutCounterStride, (uint32_t)__popcll(m));
}

// one tile entry / one padding entry of the expansion (see gut_expand_kernel)
__device__ __forceinline__ void emit_entry(bool direct, uint32_t ord_shift, uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ tile_vals,
                                           uint32_t* __restrict__ pos_particle, uint32_t slot, uint32_t ordinal, uint32_t tile, uint32_t particle) {
    if (direct) {
        tile_keys[slot] = tile | (ordinal << ord_shift);
        tile_vals[slot] = particle;
    } else {   // the sort carries the expansion position, not the particle
        tile_keys[slot] = tile;
        if (tile_vals) tile_vals[slot] = slot;
        pos_particle[slot] = particle;
    }
}
__device__ __forceinline__ void emit_padding(bool direct, uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ tile_vals,
                                             uint32_t* __restrict__ pos_particle, uint32_t q) {
    tile_keys[q] = 0xFFFFFFFFu;
    if (direct) {
        tile_vals[q] = 0xFFFFFFFFu;
    } else {
        if (tile_vals) tile_vals[q] = q;
        pos_particle[q] = 0xFFFFFFFFu;
    }
}
// ---------------------------------------------------------------------------------------------
// K4: ordered expansion — GUTProjector::expand (gutProjector.cuh:324-388), iterated in depth-rank order
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gut_expand_kernel(GutParams P, GutProjected proj, const uint32_t* __restrict__ rank_to_particle,
                                                         const uint32_t* __restrict__ offsets, uint32_t capacity,
                                                         uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ tile_vals,
                                                         uint32_t* __restrict__ pos_particle) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t off = 0, max_off = 0, p = 0;
    float cx = 0.f, cy = 0.f, pmax = 0.f;
    float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
    TileBBox bb = {0, 0, 0, 0};
    uint32_t walk_kind = 0u, walk_mask = 0u;
    const bool cached_walks = P.tile_culling != 0 && proj.walk8 != nullptr;   // (GRUT_GUT_NO_WALK_CACHE=1: the expansion tests every tile again)
    // direct lists (GutParams::rec64): key = tile | ordinal << ord_shift (the ordinal of a particle's entry is below its tile count, hence below
    // 1 << ord_shift), payload = particle; legacy: key = tile, payload = the expansion position (an iota the sort generates), pos_particle[]
    const bool direct = P.rec64 != nullptr;
    const uint32_t osh = P.ord_shift;
    if (r < P.N) {
        off = r == 0 ? 0u : offsets[r - 1];
        max_off = min(offsets[r], capacity);
        if (max_off > off) {
            p = rank_to_particle[r];
            proj.part_offset[p] = off;
            if (direct) reinterpret_cast<uint32_t*>(proj.rec64 + 4 * (size_t)p + 2)[3] = off;
            const uint2 w8 = cached_walks ? proj.walk8[p] : make_uint2(0u, 0u);
            walk_kind = w8.x >> 30;
            walk_mask = w8.y;
            if (walk_kind != 0u) {   // the counting pass's one-step walk: box and keep mask as it left them
                bb.minx = (int)(w8.x & 0xFFFu); bb.miny = (int)((w8.x >> 12) & 0xFFFu);
                bb.maxx = bb.minx + 1 + (int)((w8.x >> 24) & 7u); bb.maxy = bb.miny + 1 + (int)((w8.x >> 27) & 7u);
            } else {
                const float2 ext = proj.extent[p];
                if (!(ext.x <= 1e-06f)) {
                    const float2 c = proj.proj_pos[p];
                    cx = c.x; cy = c.y;
                    bb = tile_space_bbox(P.gx, P.gy, cx, cy, ext.x, ext.y);
                    co = proj.conic_opacity[p];
                    pmax = logf(co.w / P.min_alpha);
                }
            }
        }
    }
    // Entries of one particle land contiguously at [off, max_off); their order inside the range is irrelevant: the tile
    // sort that follows is stable per tile and each (tile, particle) pair occurs once.  Small boxes: four particles at a
    // time, one per 16-lane row; boxes up to 8x4 / 4x8: two at a time, one per half wave; large boxes: one particle at a
    // time across the wave (same split as the counting pass).
    const TileConic tc = tile_conic(co);
    const bool has = max_off > off;
    const bool small = has && bbox_fits_row(bb);
    const bool culling = P.tile_culling != 0;
    const int row_shift = lane & 48;
    const uint32_t row_lt = (1u << (lane & 15)) - 1u;
    const GroupPick rows = group_rank<16>(lane, small);
    for (int k = 0; k < rows.steps; ++k) {
        bool act;
        const int src = group_source<16>(lane, small, rows, k, act);
        const uint32_t sp = (uint32_t)row_bcast_i((int)p, src), send = act ? (uint32_t)row_bcast_i((int)max_off, src) : 0u;
        uint32_t o = (uint32_t)row_bcast_i((int)off, src);
        const uint32_t o0 = o;
        if (cached_walks) {   // (with culling on, every box that fits a row carries its mask)
            const int sminx = row_bcast_i(bb.minx, src), sminy = row_bcast_i(bb.miny, src);
            const uint32_t m = act ? (uint32_t)row_bcast_i((int)walk_mask, src) : 0u;
            const bool keep = (m >> (lane & 15)) & 1u;
            const uint32_t slot = o + (uint32_t)__popc(m & row_lt);
            if (keep && slot < send) emit_entry(direct, osh, tile_keys, tile_vals, pos_particle, slot, slot - o0, (uint32_t)((sminy + ((lane >> 2) & 3)) * P.gx + sminx + (lane & 3)), sp);
            for (uint32_t q = o + (uint32_t)__popc(m) + (lane & 15); q < send; q += 16) emit_padding(direct, tile_keys, tile_vals, pos_particle, q);
            continue;
        }
        const TileBBox sb = {row_bcast_i(bb.minx, src), row_bcast_i(bb.miny, src), row_bcast_i(bb.maxx, src), row_bcast_i(bb.maxy, src)};
        const TileConic sco = {row_bcast_f(tc.cx, src), row_bcast_f(tc.cy, src), row_bcast_f(tc.cz, src), row_bcast_f(tc.rcpx, src),
                               row_bcast_f(tc.rcpy, src)};
        row_tile_walk(lane, P.gx, culling, act, sb, sco, row_bcast_f(cx, src), row_bcast_f(cy, src), row_bcast_f(pmax, src),
                      [&](bool keep, uint32_t tile) {
                          const uint32_t m = (uint32_t)((__ballot(keep) >> row_shift) & 0xFFFFull);
                          const uint32_t slot = o + (uint32_t)__popc(m & row_lt);
                          if (keep && slot < send) emit_entry(direct, osh, tile_keys, tile_vals, pos_particle, slot, slot - o0, tile, sp);
                          o += (uint32_t)__popc(m);
                      });
        for (uint32_t q = o + (lane & 15); q < send; q += 16) emit_padding(direct, tile_keys, tile_vals, pos_particle, q);  // gutProjector.cuh:372-376 padding
    }
    const bool halfc = has && !small && bbox_fits_half(bb);
    const int half_shift = lane & 32;
    const uint32_t half_lt = (uint32_t)((1ull << (lane & 31)) - 1ull);
    const GroupPick halves = group_rank<32>(lane, halfc);
    for (int k = 0; k < halves.steps; ++k) {
        bool act;
        const int src = group_source<32>(lane, halfc, halves, k, act);
        const TileBBox sb = {row_bcast_i(bb.minx, src), row_bcast_i(bb.miny, src), row_bcast_i(bb.maxx, src), row_bcast_i(bb.maxy, src)};
        const uint32_t sp = (uint32_t)row_bcast_i((int)p, src), send = act ? (uint32_t)row_bcast_i((int)max_off, src) : 0u;
        const uint32_t o = (uint32_t)row_bcast_i((int)off, src);
        int x, y;
        bool keep = half_block_tile(lane, sb, x, y) && act;
        uint32_t m;
        if (cached_walks) {
            m = act ? (uint32_t)row_bcast_i((int)walk_mask, src) : 0u;
            keep = (m >> (lane & 31)) & 1u;
        } else {
            if (keep && culling) {
                const TileConic sco = {row_bcast_f(tc.cx, src), row_bcast_f(tc.cy, src), row_bcast_f(tc.cz, src), row_bcast_f(tc.rcpx, src),
                                       row_bcast_f(tc.rcpy, src)};
                keep = tile_min_power((float)x, (float)y, sco, row_bcast_f(cx, src), row_bcast_f(cy, src)) < row_bcast_f(pmax, src);
            }
            m = (uint32_t)((__ballot(keep) >> half_shift) & 0xFFFFFFFFull);
        }
        const uint32_t slot = o + (uint32_t)__popc(m & half_lt);
        if (keep && slot < send) emit_entry(direct, osh, tile_keys, tile_vals, pos_particle, slot, slot - o, (uint32_t)(y * P.gx + x), sp);
        for (uint32_t q = o + (uint32_t)__popc(m) + (lane & 31); q < send; q += 32) emit_padding(direct, tile_keys, tile_vals, pos_particle, q);  // gutProjector.cuh:372-376 padding
    }
    unsigned long long todo = __ballot(has && !small && !halfc);
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const TileBBox sb = {bcast_i(bb.minx, src), bcast_i(bb.miny, src), bcast_i(bb.maxx, src), bcast_i(bb.maxy, src)};
        const TileConic sco = {bcast_f(tc.cx, src), bcast_f(tc.cy, src), bcast_f(tc.cz, src), bcast_f(tc.rcpx, src), bcast_f(tc.rcpy, src)};
        const uint32_t sp = (uint32_t)bcast_i((int)p, src), send = (uint32_t)bcast_i((int)max_off, src);
        uint32_t o = (uint32_t)bcast_i((int)off, src);
        const uint32_t o0 = o;
        coop_tile_walk(lane, P.gx, culling, sb, sco, bcast_f(cx, src), bcast_f(cy, src), bcast_f(pmax, src),
                       [&](bool keep, uint32_t tile) {
                           const unsigned long long m = __ballot(keep);
                           const uint32_t slot = o + (uint32_t)__popcll(m & lt_mask);
                           if (keep && slot < send) emit_entry(direct, osh, tile_keys, tile_vals, pos_particle, slot, slot - o0, tile, sp);
                           o += (uint32_t)__popcll(m);
                       });
        for (uint32_t q = o + lane; q < send; q += 64) emit_padding(direct, tile_keys, tile_vals, pos_particle, q);
    }
}

__global__ __launch_bounds__(256) void gut_gather_particle_idx_kernel(uint32_t n, const uint32_t* __restrict__ sorted_pos,
                                                                      const uint32_t* __restrict__ pos_particle, uint32_t* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = pos_particle ? pos_particle[sorted_pos[k]] : sorted_pos[k];   // (direct lists: the payload is the particle already)
}

// ---------------------------------------------------------------------------------------------
// K6: tile ranges — computeSortedTileRangeIndices (gutRenderer.cu:46-76)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gut_tile_ranges_kernel(uint32_t n_cap, const uint32_t* __restrict__ n_dev, uint32_t tile_mask,
                                                              uint32_t num_tiles, const uint32_t* __restrict__ sorted_tile_keys,
                                                              uint2* __restrict__ ranges, uint32_t* __restrict__ boundary_tile) {
    // four consecutive keys per thread (one 16-byte load + the key in front of them)
    const uint32_t k0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    const uint32_t n = n_dev ? min(n_cap, *n_dev) : n_cap;  // n_cap may be a capacity bound (speculative launch)
    if (k0 >= n) return;
    uint32_t key[4];
    if (k0 + 4u <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(sorted_tile_keys + k0);
        key[0] = v.x; key[1] = v.y; key[2] = v.z; key[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) key[j] = (k0 + j < n) ? sorted_tile_keys[k0 + j] : 0u;
    }
    uint32_t pt = k0 ? (sorted_tile_keys[k0 - 1] & tile_mask) : 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t k = k0 + j;
        if (k >= n) break;
        const uint32_t t = key[j] & tile_mask;
        const bool valid = t < num_tiles;
        // segment boundary b sits at sorted index b * kGutSegment; remember which tile's list it cuts
        if ((k % kGutSegment) == 0) boundary_tile[k / kGutSegment] = valid ? t : 0xFFFFFFFFu;
        if (k == 0) {
            if (valid) ranges[t].x = 0;
        } else if (pt != t) {
            if (pt < num_tiles) ranges[pt].y = k;
            if (valid) ranges[t].x = k;
        }
        if (valid && k == n - 1) ranges[t].y = n;
        pt = t;
    }
}

// ---------------------------------------------------------------------------------------------
// K9: gradient finalisation = per-particle gather of the compositing partials + projection backward
// (GUTProjector::evalBackward, gutProjector.cuh:390-430).  One lane per particle:
//   * sums the particle's slots [2 off, 2 (off + count)) that the gradient sweep flagged (particles covering many
//     tiles are gathered by the whole wave instead, so that no lane becomes the critical path);
//   * contracts (B, M) into the position / quaternion / scale gradients (matmul_bw_vec, matmul_bw_quat and the
//     1/scale chain of models/gaussianParticles.cuh:624-751, applied once per particle instead of once per hit);
//   * pushes the radiance gradient through the clamp and the SH basis (dRGB -> dSH, d direction -> d position);
//   * writes the complete [N,12] and [N,3*ncoef] gradient rows (zeros for particles without tiles): nothing is
//     accumulated in place, so the result does not depend on what the output buffers held.
// SH rows travel through LDS (row stride 49 floats: conflict-free both ways): a wave reads the coefficient rows of
// its 64 particles and writes their gradient rows as contiguous 192-byte segments instead of 64 scattered rows.
// ---------------------------------------------------------------------------------------------
// tile counts up to this are gathered by the owning quad, larger ones by the whole wave.  A wave of 16 quads runs as long as
// its largest particle (mean 17 tiles, p99 84, max 743 on the bench cloud: 23 % efficiency if every quad gathers alone);
// gather + projection backward at thresholds 128 / 64 / 32 / 16: 0.262 / 0.241 / 0.227 / 0.227 ms
constexpr uint32_t kGatherSmall = 32;
struct __attribute__((packed, aligned(2))) FlagChunk {
    unsigned long long lo, hi;
};

// gradient of sum_ij m_ij rotT_ij(q) w.r.t. q = (r,x,y,z); q2 = 2q (matmul_bw_quat, mathUtils.cuh:458-521)
__device__ __forceinline__ float4 quat_contract(const float m[9], float4 q2) {
    const float r = q2.x, x = q2.y, y = q2.z, z = q2.w;
    const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    // rotT = [[1-2(yy+zz), 2(xy+rz), 2(xz-ry)], [2(xy-rz), 1-2(xx+zz), 2(yz+rx)], [2(xz+ry), 2(yz-rx), 1-2(xx+yy)]]
    const float s01 = m01 + m10, s02 = m02 + m20, s12 = m12 + m21;   // symmetric parts
    const float a01 = m01 - m10, a02 = m20 - m02, a12 = m12 - m21;   // antisymmetric parts (signs as in rotT)
    float4 d;
    d.x = z * a01 + y * a02 + x * a12;
    d.y = y * s01 + z * s02 + r * a12 - 2.f * x * (m11 + m22);
    d.z = x * s01 + z * s12 + r * a02 - 2.f * y * (m00 + m22);
    d.w = x * s02 + y * s12 + r * a01 - 2.f * z * (m00 + m11);
    return d;
}

// Four lanes per particle: lane c of a quad owns column c (one float4) of the 16-float slot rows, so that a row is ONE
// 64-byte request of four neighbouring lanes (a lane per particle needed four requests per row, each touching a different
// cache line in every lane: the texture path, not HBM, was the limit).  With hit-distance gradients the rows have a fifth
// column, read by lane 0 of the quad.
template <int STRIDE>
struct QuadAcc {
    float4 a;
    float4 x;  // fifth column (STRIDE == 20), lane c == 0 only
    __device__ __forceinline__ void clear() { a = make_float4(0.f, 0.f, 0.f, 0.f); x = a; }
    // rows sb+b0 .. sb+b3 (b < 0: none), requested together and summed in this order
    __device__ __forceinline__ void add_rows4(const float* __restrict__ partial, size_t sb, int b0, int b1, int b2, int b3, int c) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* r0 = reinterpret_cast<const float4*>(partial + (sb + (size_t)b0) * STRIDE);
        const float4* r1 = reinterpret_cast<const float4*>(partial + (sb + (size_t)(b1 < 0 ? b0 : b1)) * STRIDE);
        const float4* r2 = reinterpret_cast<const float4*>(partial + (sb + (size_t)(b2 < 0 ? b0 : b2)) * STRIDE);
        const float4* r3 = reinterpret_cast<const float4*>(partial + (sb + (size_t)(b3 < 0 ? b0 : b3)) * STRIDE);
        const float4 v0 = r0[c], v1 = r1[c], v2 = r2[c], v3 = r3[c];
        float4 w0 = z, w1 = z, w2 = z, w3 = z;
        if (STRIDE > 16 && c == 0) { w0 = r0[4]; w1 = r1[4]; w2 = r2[4]; w3 = r3[4]; }
        a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
        if (b1 >= 0) { a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w; }
        if (b2 >= 0) { a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w; }
        if (b3 >= 0) { a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w; }
        if (STRIDE > 16 && c == 0) {
            x.x += w0.x; x.y += w0.y; x.z += w0.z; x.w += w0.w;
            if (b1 >= 0) { x.x += w1.x; x.y += w1.y; x.z += w1.z; x.w += w1.w; }
            if (b2 >= 0) { x.x += w2.x; x.y += w2.y; x.z += w2.z; x.w += w2.w; }
            if (b3 >= 0) { x.x += w3.x; x.y += w3.y; x.z += w3.z; x.w += w3.w; }
        }
    }
    __device__ __forceinline__ void add_row(const float* __restrict__ partial, size_t slot, int c) {
        const float4* row = reinterpret_cast<const float4*>(partial + slot * STRIDE);
        const float4 v = row[c];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        if (STRIDE > 16 && c == 0) {
            const float4 w = row[4];
            x.x += w.x; x.y += w.y; x.z += w.z; x.w += w.w;
        }
    }
};
template <int J>
__device__ __forceinline__ float quad_bcast(float v) {  // value of lane J of the caller's quad
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, false));
}
template <int X>
__device__ __forceinline__ float quad_bcast_xor(float v) {  // value of lane (own ^ X) of the caller's quad, X = 1 or 2
    constexpr int perm = X == 1 ? (1 | (0 << 2) | (3 << 4) | (2 << 6)) : (2 | (3 << 2) | (0 << 4) | (1 << 6));
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), perm, 0xf, 0xf, false));
}
__device__ __forceinline__ float xor_sum_quads(float v) {  // sum over the 16 quads of a wave, same column: lanes c, c+4, ...
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// NHT (neural harmonic features, gut_render_nht.inl): slot words 13..15 carry the direct hit-distance scale terms instead of a radiance
// gradient, and nothing is handed on to the SH backward.
// FUSE_SH (round 5; fp32 rows of 48 coefficients, the expanded - not factored - SH gradient): the projection backward runs in the same
// kernel.  The quad that gathered a particle splits its 16 SH coefficients four ways: lane c owns coefficients 4c .. 4c+3 = floats
// [12 c, 12 c + 12) of the row, which are three consecutive 16-byte words - so the wave's 16 rows are read and their gradient rows written
// as 3 float4 per lane at float4 index 3 lane + {0, 1, 2}, straight from / to global memory (no LDS transpose), requested at the top of the
// kernel so that their round trip lies under the flag -> row chain.  Rows of particles without tiles are not read (a third of them) and
// their gradient rows are written as zeros.  Saves a launch, the g_rgb round trip, the second read of the particle rows and the
// read-modify-write of the position gradient.
template <int STRIDE, bool NHT = false, bool FUSE_SH = false>
__global__ __launch_bounds__(256) void gut_grad_gather_kernel(GutParams P, GutProjected proj, const float4* __restrict__ density12,
                                                              GutGradSlots slots, int have_partials, GutGradOut g_out,
                                                              float* __restrict__ g_rgb, uint32_t first, uint32_t end,
                                                              const float* __restrict__ sph = nullptr, float* __restrict__ g_sph = nullptr) {
    // particles [first, end): the whole scene, or one chunk of the pipelined gradient exchange (gut_backward_factored_chunked)
    const int lane = threadIdx.x & 63, c = threadIdx.x & 3;
    const uint32_t i = first + blockIdx.x * 64u + (threadIdx.x >> 2);
    const uint32_t count = (i < end) ? proj.tiles_count[i] : 0u;
    const bool has = count != 0;
    QuadAcc<STRIDE> acc;
    acc.clear();
    const uint32_t off = has ? proj.part_offset[i] : 0u;
    // the particle's own parameters are needed only by the closing arithmetic; requested here, their round trip overlaps the
    // flag -> row chain below instead of following it
    float4 q = make_float4(1.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 0.f);
    if (has) { q = density12[3 * (size_t)i + 1]; sc = density12[3 * (size_t)i + 2]; }
    float4 sh0 = make_float4(0.f, 0.f, 0.f, 0.f), sh1 = sh0, sh2 = sh0, pa = sh0;
    f3 rad = mk3(0.f, 0.f, 0.f);
    if (FUSE_SH && has) {
        const float4* row = reinterpret_cast<const float4*>(sph + 48 * (size_t)i) + 3 * c;
        sh0 = row[0]; sh1 = row[1]; sh2 = row[2];
        pa = density12[3 * (size_t)i];
        rad = mk3(proj.rgb[3 * (size_t)i], proj.rgb[3 * (size_t)i + 1], proj.rgb[3 * (size_t)i + 2]);
    }
    if (have_partials) {
        if (has && count <= kGatherSmall) {
            // Set flags are rare (most tile entries lie behind the rays' termination) and every load here is a dependent
            // round trip to HBM, so requests are issued in groups: 64 flags (four 16-byte loads) are packed into one bit
            // mask, then the rows of up to four set bits are requested together before any of them is summed.
            const size_t s0 = 2 * (size_t)off;
            const uint32_t nb = 2 * count;
            for (uint32_t base = 0; base < nb; base += 64) {
                const FlagChunk* fc = reinterpret_cast<const FlagChunk*>(slots.flag + s0 + base);  // buffer has 96 B of slack
                const FlagChunk w0 = fc[0], w1 = fc[1], w2 = fc[2], w3 = fc[3];
                // gather bit 0 of each of 8 flag bytes into one byte (bytes past the flag array are arbitrary: keep bit 0 only,
                // or they would carry into their neighbours' bits)
                constexpr unsigned long long kPack = 0x0102040810204080ull, kLsb = 0x0101010101010101ull;
                auto pack8 = [](unsigned long long w) { return ((w & kLsb) * kPack) >> 56; };
                unsigned long long m = pack8(w0.lo) | (pack8(w0.hi) << 8) | (pack8(w1.lo) << 16) | (pack8(w1.hi) << 24) |
                                       (pack8(w2.lo) << 32) | (pack8(w2.hi) << 40) | (pack8(w3.lo) << 48) | (pack8(w3.hi) << 56);
                const uint32_t rem = nb - base;
                if (rem < 64) m &= (1ull << rem) - 1ull;
                const size_t sb = s0 + base;
                while (m) {
                    const int b0 = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int b1 = m ? __ffsll((long long)m) - 1 : -1;
                    m &= m - 1;   // m == 0 stays 0
                    const int b2 = m ? __ffsll((long long)m) - 1 : -1;
                    m &= m - 1;
                    const int b3 = m ? __ffsll((long long)m) - 1 : -1;
                    m &= m - 1;
                    acc.add_rows4(slots.partial, sb, b0, b1, b2, b3, c);
                }
            }
        }
        // the heavy tail: all 16 quads of the wave share one particle's slots
        unsigned long long big = __ballot(has && count > kGatherSmall && c == 0);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const size_t s0 = 2 * (size_t)(uint32_t)__builtin_amdgcn_readlane((int)off, src);
            const uint32_t n2 = 2u * (uint32_t)__builtin_amdgcn_readlane((int)count, src);
            QuadAcc<STRIDE> part;
            part.clear();
            // 64 flags per step, one byte per lane, turned into a wave-uniform mask by a ballot; quad g owns bits g, g+16, g+32,
            // g+48 of it and requests its set rows together
            const int g = lane >> 2;
            for (uint32_t base = 0; base < n2; base += 64) {
                const bool set = (base + (uint32_t)lane < n2) && slots.flag[s0 + base + lane] != 0;
                const unsigned long long m = __ballot(set);
                if (m == 0) continue;
                int b[4], nb = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((m >> (g + 16 * j)) & 1ull) b[nb++] = g + 16 * j;
                if (nb > 0)
                    part.add_rows4(slots.partial, s0 + base, b[0], nb > 1 ? b[1] : -1, nb > 2 ? b[2] : -1, nb > 3 ? b[3] : -1, c);
            }
            part.a.x = xor_sum_quads(part.a.x); part.a.y = xor_sum_quads(part.a.y);
            part.a.z = xor_sum_quads(part.a.z); part.a.w = xor_sum_quads(part.a.w);
            if (STRIDE > 16) {
                part.x.x = xor_sum_quads(part.x.x); part.x.y = xor_sum_quads(part.x.y);
                part.x.z = xor_sum_quads(part.x.z); part.x.w = xor_sum_quads(part.x.w);
            }
            if ((lane >> 2) == (src >> 2)) acc = part;
        }
    }
    // every lane of the quad collects the whole row sum (column j from lane j) and redoes the small closing arithmetic;
    // lane c then stores output column c, so a particle's 48 + 12 bytes leave as neighbouring requests
    float r[20];
    r[0] = quad_bcast<0>(acc.a.x); r[1] = quad_bcast<0>(acc.a.y); r[2] = quad_bcast<0>(acc.a.z); r[3] = quad_bcast<0>(acc.a.w);
    r[4] = quad_bcast<1>(acc.a.x); r[5] = quad_bcast<1>(acc.a.y); r[6] = quad_bcast<1>(acc.a.z); r[7] = quad_bcast<1>(acc.a.w);
    r[8] = quad_bcast<2>(acc.a.x); r[9] = quad_bcast<2>(acc.a.y); r[10] = quad_bcast<2>(acc.a.z); r[11] = quad_bcast<2>(acc.a.w);
    r[12] = quad_bcast<3>(acc.a.x); r[13] = quad_bcast<3>(acc.a.y); r[14] = quad_bcast<3>(acc.a.z); r[15] = quad_bcast<3>(acc.a.w);
    r[16] = quad_bcast<0>(acc.x.x); r[17] = quad_bcast<0>(acc.x.y); r[18] = quad_bcast<0>(acc.x.z); r[19] = 0.f;
    if (i >= end) return;
    float4* gd = reinterpret_cast<float4*>(g_out.packed + 12 * (size_t)i);
    if (FUSE_SH && !has) {
        float4* out = reinterpret_cast<float4*>(g_sph + 48 * (size_t)i) + 3 * c;
        out[0] = out[1] = out[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!has) {
        if (g_out.packed) {
            if (c < 3) gd[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (c == 0) {
            g_out.pos[3 * (size_t)i] = 0.f; g_out.pos[3 * (size_t)i + 1] = 0.f; g_out.pos[3 * (size_t)i + 2] = 0.f;
            g_out.dns[i] = 0.f;
        } else if (c == 1) {
            reinterpret_cast<float4*>(g_out.rot)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (c == 2) {
            g_out.scl[3 * (size_t)i] = 0.f; g_out.scl[3 * (size_t)i + 1] = 0.f; g_out.scl[3 * (size_t)i + 2] = 0.f;
        }
        return;
    }
    const m3 rt = quat_wxyz_to_rotT(q.x, q.y, q.z, q.w);
    const f3 B = mk3(r[0], r[1], r[2]);
    const float* m = &r[4];
    // position = -R B  (matmul_bw_vec with rows of R^T); the SH direction term is added by the projection backward
    const f3 gpos = mk3(-(B.x * rt.r0.x + B.y * rt.r1.x + B.z * rt.r2.x), -(B.x * rt.r0.y + B.y * rt.r1.y + B.z * rt.r2.y),
                        -(B.x * rt.r0.z + B.y * rt.r1.z + B.z * rt.r2.z));
    // scale_i = -(1/s_i) sum_j rotT_ij m_ij (+ direct hit-distance term)
    float gsx = -(rt.r0.x * m[0] + rt.r0.y * m[1] + rt.r0.z * m[2]) / sc.x;
    float gsy = -(rt.r1.x * m[3] + rt.r1.y * m[4] + rt.r1.z * m[5]) / sc.y;
    float gsz = -(rt.r2.x * m[6] + rt.r2.y * m[7] + rt.r2.z * m[8]) / sc.z;
    if (STRIDE > 16) { gsx += r[16]; gsy += r[17]; gsz += r[18]; }
    if (NHT) { gsx += r[13]; gsy += r[14]; gsz += r[15]; }
    const float4 dq = quat_contract(m, make_float4(2.f * q.x, 2.f * q.y, 2.f * q.z, 2.f * q.w));
    f3 gpos_out = gpos;
    if (FUSE_SH) {   // GUTProjector::evalBackward (gutProjector.cuh:390-430): dRGB -> dSH through the clamp, d direction -> d position
        const FramePoses& FP = frame_poses(P);
        const int nact = min((P.n_active + 1) * (P.n_active + 1), P.ncoef);
        float ilen;
        const f3 dir = sh_view_direction(mk3(pa.x, pa.y, pa.z), mk3(FP.s2w_t[0], FP.s2w_t[1], FP.s2w_t[2]), ilen);
        f3 g = mk3(r[13], r[14], r[15]);
        if (!(rad.x > 0.f)) g.x = 0.f;     // clamp mask on the unclamped radiance stored by the forward projection
        if (!(rad.y > 0.f)) g.y = 0.f;
        if (!(rad.z > 0.f)) g.z = 0.f;
        const float rowv[12] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w, sh2.x, sh2.y, sh2.z, sh2.w};
        float o[12];
        f3 gdir = mk3(0.f, 0.f, 0.f);
        // the lane's four coefficients k = 4 c + j: one branch per c with static indices (a select over c is turned into a dynamically
        // indexed array by the compiler - 208 B of scratch); each branch keeps only its own four basis functions alive
        auto part = [&](auto group) {
            constexpr int G = decltype(group)::value;
            float basis[16];
            f3 dbasis[16];
            sh_basis_pinned(P.n_active, dir, basis);
            sh_basis_grad_pinned(P.n_active, dir, dbasis);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = 4 * G + j < nact;
                o[3 * j] = o[3 * j + 1] = o[3 * j + 2] = 0.f;
                if (on) {   // (inactive degrees: their basis functions are not evaluated - nothing of them may enter a sum)
                    const float s = sh_row_dot(g, rowv[3 * j], rowv[3 * j + 1], rowv[3 * j + 2]);
                    gdir = sh_axpy(gdir, dbasis[4 * G + j], s);
                    o[3 * j] = mul_rn(basis[4 * G + j], g.x); o[3 * j + 1] = mul_rn(basis[4 * G + j], g.y); o[3 * j + 2] = mul_rn(basis[4 * G + j], g.z);
                }
            }
        };
        switch (c) {
        case 0: part(std::integral_constant<int, 0>{}); break;
        case 1: part(std::integral_constant<int, 1>{}); break;
        case 2: part(std::integral_constant<int, 2>{}); break;
        default: part(std::integral_constant<int, 3>{}); break;
        }
        // sum of the four lanes' parts (quad butterflies)
        gdir = sh_sum(gdir, mk3(quad_bcast_xor<1>(gdir.x), quad_bcast_xor<1>(gdir.y), quad_bcast_xor<1>(gdir.z)));   // (p0 + p1), (p2 + p3)
        gdir = sh_sum(gdir, mk3(quad_bcast_xor<2>(gdir.x), quad_bcast_xor<2>(gdir.y), quad_bcast_xor<2>(gdir.z)));   // their sum
        const f3 gsh = sh_position_gradient(dir, ilen, gdir);
        gpos_out = mk3(add_rn(gpos.x, gsh.x), add_rn(gpos.y, gsh.y), add_rn(gpos.z, gsh.z));
        float4* out = reinterpret_cast<float4*>(g_sph + 48 * (size_t)i) + 3 * c;
        out[0] = make_float4(o[0], o[1], o[2], o[3]);
        out[1] = make_float4(o[4], o[5], o[6], o[7]);
        out[2] = make_float4(o[8], o[9], o[10], o[11]);
    }
    if (FUSE_SH) {
        if (c == 0) {
            if (g_out.packed) gd[0] = make_float4(gpos_out.x, gpos_out.y, gpos_out.z, r[3]);
            else {
                g_out.pos[3 * (size_t)i] = gpos_out.x; g_out.pos[3 * (size_t)i + 1] = gpos_out.y; g_out.pos[3 * (size_t)i + 2] = gpos_out.z;
                g_out.dns[i] = r[3];
            }
        } else if (c == 1) {
            if (g_out.packed) gd[1] = dq;
            else reinterpret_cast<float4*>(g_out.rot)[i] = dq;
        } else if (c == 2) {
            if (g_out.packed) gd[2] = make_float4(gsx, gsy, gsz, 0.f);
            else { g_out.scl[3 * (size_t)i] = gsx; g_out.scl[3 * (size_t)i + 1] = gsy; g_out.scl[3 * (size_t)i + 2] = gsz; }
        }
        return;
    }
    if (c == 0) {
        if (g_out.packed) gd[0] = make_float4(gpos.x, gpos.y, gpos.z, r[3]);
        else {
            g_out.pos[3 * (size_t)i] = gpos.x; g_out.pos[3 * (size_t)i + 1] = gpos.y; g_out.pos[3 * (size_t)i + 2] = gpos.z;
            g_out.dns[i] = r[3];
        }
    } else if (c == 1) {
        if (g_out.packed) gd[1] = dq;
        else reinterpret_cast<float4*>(g_out.rot)[i] = dq;
    } else if (c == 2) {
        if (g_out.packed) gd[2] = make_float4(gsx, gsy, gsz, 0.f);
        else { g_out.scl[3 * (size_t)i] = gsx; g_out.scl[3 * (size_t)i + 1] = gsy; g_out.scl[3 * (size_t)i + 2] = gsz; }
    } else if (!NHT) {
        g_rgb[3 * (size_t)i] = r[13];
        g_rgb[3 * (size_t)i + 1] = r[14];
        g_rgb[3 * (size_t)i + 2] = r[15];
    }
}

constexpr int kShStride = 49;
// FACTORED (view-sharded data parallelism, gut_backward_factored): the SH-coefficient gradient basis_k(dir) x g is NOT
// expanded here (192 B per particle); the view-specific factor g — the radiance gradient behind the clamp — is written to
// g_radiance [N+1,3] instead, with the sensor position of the view in row N, and grut_sph_grad_from_views rebuilds the sum
// over views after the factors have been gathered.  The view-direction part of the position gradient stays here.
template <bool FACTORED>
__global__ __launch_bounds__(128) void gut_project_bwd_kernel(GutParams P, const uint32_t* __restrict__ tiles_count,
                                                              const float4* __restrict__ density12, const float* __restrict__ sph,
                                                              const float* __restrict__ rgb, const float* __restrict__ g_rgb,
                                                              GutGradOut g_out, float* __restrict__ g_sph,
                                                              float* __restrict__ g_radiance, uint32_t first, uint32_t end) {
    __shared__ float s_rows[2][64 * kShStride];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t wave_base = first + blockIdx.x * 128u + (uint32_t)wave * 64u;   // particles [first, end), first a multiple of 128
    const uint32_t i = wave_base + lane;
    const int rowlen = 3 * P.ncoef;
    const int nact = min((P.n_active + 1) * (P.n_active + 1), P.ncoef);
    float* rows = s_rows[wave];
    const FramePoses& FP = frame_poses(P);
    const bool has = (i < end) && (tiles_count[i] != 0);
    const unsigned long long has_mask = __ballot(has);
    const int nrows = (int)min(64u, end > wave_base ? end - wave_base : 0u);
    // stage in.  48-float rows: the wave's rows are one contiguous, 16-byte aligned block, copied as independent float4
    // requests (12 per lane in flight); rows of particles without tiles ride along.  Other row lengths: one row per step.
    if (P.sph_half) {
        const __half* hs = reinterpret_cast<const __half*>(sph);
        for (unsigned long long m = has_mask; m;) {
            const int row = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (lane < 3 * nact) rows[row * kShStride + lane] = __half2float(hs[(size_t)(wave_base + row) * rowlen + lane]);
        }
    } else if (rowlen == 48) {
        if (has_mask) {
            const float4* src = reinterpret_cast<const float4*>(sph + (size_t)wave_base * 48);
            const int total4 = nrows * 12;
#pragma unroll
            for (int it = 0; it < 12; ++it) {
                const int q = lane + 64 * it;
                if (q < total4) {
                    const float4 v = src[q];
                    float* d = rows + (q / 12) * kShStride + 4 * (q % 12);
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                }
            }
        }
    } else {
        for (unsigned long long m = has_mask; m;) {
            const int row = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (lane < 3 * nact) rows[row * kShStride + lane] = sph[(size_t)(wave_base + row) * rowlen + lane];
        }
    }
    __syncthreads();
    float* myrow = rows + lane * kShStride;
    if (has) {
        const float4 a = density12[3 * (size_t)i];
        float ilen;
        const f3 dir = sh_view_direction(mk3(a.x, a.y, a.z), mk3(FP.s2w_t[0], FP.s2w_t[1], FP.s2w_t[2]), ilen);
        f3 g = mk3(g_rgb[3 * (size_t)i], g_rgb[3 * (size_t)i + 1], g_rgb[3 * (size_t)i + 2]);
        // clamp mask on the unclamped radiance stored by the forward projection
        if (!(rgb[3 * (size_t)i] > 0.f)) g.x = 0.f;
        if (!(rgb[3 * (size_t)i + 1] > 0.f)) g.y = 0.f;
        if (!(rgb[3 * (size_t)i + 2] > 0.f)) g.z = 0.f;
        float basis[16];
        f3 dbasis[16];
        sh_basis_pinned(P.n_active, dir, basis);
        sh_basis_grad_pinned(P.n_active, dir, dbasis);
        // summed in four groups of four coefficients, then (p0 + p1) + (p2 + p3): the order of the fused gather kernel, whose quads split
        // the coefficients four ways (FUSE_SH above) - every path to the position gradient gives the same bits
        f3 part[4];
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            f3 gdir = mk3(0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * G + j;
                if (k < nact) {
                    const float s = sh_row_dot(g, myrow[3 * k], myrow[3 * k + 1], myrow[3 * k + 2]);
                    gdir = sh_axpy(gdir, dbasis[k], s);
                    if (!FACTORED) { myrow[3 * k] = mul_rn(basis[k], g.x); myrow[3 * k + 1] = mul_rn(basis[k], g.y); myrow[3 * k + 2] = mul_rn(basis[k], g.z); }
                }
            }
            part[G] = gdir;
        }
        const f3 gdir = sh_sum(sh_sum(part[0], part[1]), sh_sum(part[2], part[3]));
        if (!FACTORED)
            for (int k = 3 * nact; k < rowlen; ++k) myrow[k] = 0.f;
        const f3 gpos = sh_position_gradient(dir, ilen, gdir);
        float* gp = g_out.packed ? g_out.packed + 12 * (size_t)i : g_out.pos + 3 * (size_t)i;
        gp[0] += gpos.x;
        gp[1] += gpos.y;
        gp[2] += gpos.z;
        if (FACTORED) { g_radiance[3 * (size_t)i] = g.x; g_radiance[3 * (size_t)i + 1] = g.y; g_radiance[3 * (size_t)i + 2] = g.z; }
    } else if (!FACTORED) {
        for (int k = 0; k < rowlen; ++k) myrow[k] = 0.f;
    } else if (i < end) {
        g_radiance[3 * (size_t)i] = 0.f; g_radiance[3 * (size_t)i + 1] = 0.f; g_radiance[3 * (size_t)i + 2] = 0.f;
    }
    if (FACTORED) {
        if (first == 0u && blockIdx.x == 0 && threadIdx.x == 0) {   // the view's sensor position rides along as row N
            g_radiance[3 * (size_t)P.N] = FP.s2w_t[0]; g_radiance[3 * (size_t)P.N + 1] = FP.s2w_t[1]; g_radiance[3 * (size_t)P.N + 2] = FP.s2w_t[2];
        }
        return;
    }
    __syncthreads();
    // stage out: every row of the wave, zeros included (the caller does not pre-fill grad_sph)
    if (rowlen == 48) {
        float4* dst = reinterpret_cast<float4*>(g_sph + (size_t)wave_base * 48);
        const int total4 = nrows * 12;
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const int q = lane + 64 * it;
            if (q < total4) {
                const float* d = rows + (q / 12) * kShStride + 4 * (q % 12);
                dst[q] = make_float4(d[0], d[1], d[2], d[3]);
            }
        }
    } else {
        for (int row = 0; row < nrows; ++row)
            if (lane < rowlen) g_sph[(size_t)(wave_base + row) * rowlen + lane] = rows[row * kShStride + lane];
    }
}

// Sum over views of the SH-coefficient gradients from the gathered per-view factors (see gut_project_bwd_kernel<true>):
//   g_sph[p][k] = scale * sum_v basis_k(normalize(mu_p - cam_v)) * g_v[p]
// factors [n_views][N+1][3] (row N of each view = its sensor position).  Views are added in index order on every rank, so all
// replicas end up with bitwise identical gradients.  Rows leave through LDS as 16-byte requests, as in the kernel above.
__global__ __launch_bounds__(128) void sph_grad_from_views_kernel(uint32_t N, uint32_t n_views, const float* __restrict__ factors,
                                                                  const float* __restrict__ positions, uint32_t pos_stride, int n_active,
                                                                  int ncoef, float scale, float* __restrict__ g_sph) {
    __shared__ float s_rows[2][64 * kShStride];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t wave_base = blockIdx.x * 128u + (uint32_t)wave * 64u;
    const uint32_t i = wave_base + lane;
    const int rowlen = 3 * ncoef;
    const int nact = min((n_active + 1) * (n_active + 1), ncoef);
    float* rows = s_rows[wave];
    float* myrow = rows + lane * kShStride;
    const int nrows = (int)min(64u, N > wave_base ? N - wave_base : 0u);
    if (i < N) {
        const f3 mu = mk3(positions[(size_t)i * pos_stride], positions[(size_t)i * pos_stride + 1], positions[(size_t)i * pos_stride + 2]);
        float acc[48];
#pragma unroll
        for (int k = 0; k < 48; ++k) acc[k] = 0.f;
        for (uint32_t v = 0; v < n_views; ++v) {
            const float* view = factors + (size_t)v * 3 * ((size_t)N + 1);
            const f3 g = mk3(view[3 * (size_t)i], view[3 * (size_t)i + 1], view[3 * (size_t)i + 2]);
            if (g.x == 0.f && g.y == 0.f && g.z == 0.f) continue;   // invisible or fully clamped in this view
            float ilen;
            const f3 dir = sh_view_direction(mu, mk3(view[3 * (size_t)N], view[3 * (size_t)N + 1], view[3 * (size_t)N + 2]), ilen);
            float basis[16];
            sh_basis_pinned(n_active, dir, basis);
#pragma unroll
            for (int k = 0; k < 16; ++k)   // (product rounded on its own, then added: one view gives the single-GPU kernels' bits)
                if (k < nact) { acc[3 * k] = add_rn(acc[3 * k], mul_rn(basis[k], g.x)); acc[3 * k + 1] = add_rn(acc[3 * k + 1], mul_rn(basis[k], g.y)); acc[3 * k + 2] = add_rn(acc[3 * k + 2], mul_rn(basis[k], g.z)); }
        }
#pragma unroll
        for (int k = 0; k < 48; ++k)
            if (k < rowlen) myrow[k] = acc[k] * scale;
    }
    __syncthreads();
    if (rowlen == 48) {
        float4* dst = reinterpret_cast<float4*>(g_sph + (size_t)wave_base * 48);
        const int total4 = nrows * 12;
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const int q = lane + 64 * it;
            if (q < total4) {
                const float* d = rows + (q / 12) * kShStride + 4 * (q % 12);
                dst[q] = make_float4(d[0], d[1], d[2], d[3]);
            }
        }
    } else {
        for (int row = 0; row < nrows; ++row)
            if (lane < rowlen) g_sph[(size_t)(wave_base + row) * rowlen + lane] = rows[row * kShStride + lane];
    }
}

// Between the scan and the tail of the frame: publishes the intersection count I and the visible-particle count straight into
// the host's pinned counters (device-visible memory: no blit copies on the stream), re-arms the visible counter for the next
// frame, and clears the two small per-frame tables of the tail (tile ranges, checkpoint "reached" marks) — one launch instead of
// three fills and two copies.
__global__ __launch_bounds__(256) void gut_prepare_tail_kernel(const uint32_t* __restrict__ last_offset, uint32_t* __restrict__ num_visible,
                                                               uint32_t* __restrict__ host_counters, uint32_t* __restrict__ ranges_words,
                                                               uint32_t n_ranges_words, uint32_t* __restrict__ reached_words, uint32_t n_reached_words) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (blockIdx.x == 0 && threadIdx.x < 64) {   // one wave: the counter's replicas summed and re-armed
        uint32_t v = 0u;
        if (threadIdx.x < kGutCounterReplicas) { v = num_visible[threadIdx.x * kGutCounterStride]; num_visible[threadIdx.x * kGutCounterStride] = 0u; }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (threadIdx.x == 0) {
            __hip_atomic_store(&host_counters[0], *last_offset, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_counters[1], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    for (uint32_t k = i; k < n_ranges_words; k += stride) ranges_words[k] = 0u;
    for (uint32_t k = i; k < n_reached_words; k += stride) reached_words[k] = 0u;
}

}  // namespace

void launch_prepare_tail(hipStream_t s, const uint32_t* last_offset, uint32_t* num_visible, uint32_t* host_counters, uint32_t* ranges_words,
                         uint32_t n_ranges_words, uint32_t* reached_words, uint32_t n_reached_words) {
    hipLaunchKernelGGL(gut_prepare_tail_kernel, dim3(64), dim3(256), 0, s, last_offset, num_visible, host_counters, ranges_words, n_ranges_words,
                       reached_words, n_reached_words);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void launch_project(hipStream_t s, const GutParams& P, const float* density12, const float* sph, const GutProjected& out,
                    int32_t* visibility, uint32_t* num_visible) {
#define GRUT_PROJECT_LAUNCH(M_, R_)                                                                                                          \
    hipLaunchKernelGGL((gut_project_kernel<M_, R_>), dim3(div_up(P.N, 256)), dim3(256), 0, s, P, reinterpret_cast<const float4*>(density12), \
                       sph, out, visibility, num_visible)
    const bool rolling = P.cam.shutter != GRUT_SHUTTER_GLOBAL;
    switch (P.cam.model) {
    case GRUT_CAMERA_OPENCV_PINHOLE: if (rolling) GRUT_PROJECT_LAUNCH(GRUT_CAMERA_OPENCV_PINHOLE, 1); else GRUT_PROJECT_LAUNCH(GRUT_CAMERA_OPENCV_PINHOLE, 0); break;
    case GRUT_CAMERA_OPENCV_FISHEYE: if (rolling) GRUT_PROJECT_LAUNCH(GRUT_CAMERA_OPENCV_FISHEYE, 1); else GRUT_PROJECT_LAUNCH(GRUT_CAMERA_OPENCV_FISHEYE, 0); break;
    case GRUT_CAMERA_FTHETA: if (rolling) GRUT_PROJECT_LAUNCH(GRUT_CAMERA_FTHETA, 1); else GRUT_PROJECT_LAUNCH(GRUT_CAMERA_FTHETA, 0); break;
    default: GRUT_PROJECT_LAUNCH(-1, 1); break;
    }
#undef GRUT_PROJECT_LAUNCH
}
void launch_expand(hipStream_t s, const GutParams& P, const GutProjected& proj, const uint32_t* rank_to_particle,
                   const uint32_t* offsets, uint32_t capacity, uint32_t* tile_keys, uint32_t* tile_vals, uint32_t* pos_particle) {
    hipLaunchKernelGGL(gut_expand_kernel, dim3(div_up(P.N, 256)), dim3(256), 0, s, P, proj, rank_to_particle, offsets, capacity,
                       tile_keys, tile_vals, pos_particle);
}
void launch_gather_particle_idx(hipStream_t s, uint32_t n, const uint32_t* sorted_pos, const uint32_t* pos_particle, uint32_t* out) {
    if (n) hipLaunchKernelGGL(gut_gather_particle_idx_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, sorted_pos, pos_particle, out);
}
void launch_tile_ranges(hipStream_t s, uint32_t n, const uint32_t* n_dev, uint32_t tile_mask, uint32_t num_tiles,
                        const uint32_t* sorted_tile_keys, uint32_t* ranges, uint32_t* boundary_tile) {
    hipLaunchKernelGGL(gut_tile_ranges_kernel, dim3(div_up(div_up(n, 4u), 256)), dim3(256), 0, s, n, n_dev, tile_mask, num_tiles, sorted_tile_keys,
                       reinterpret_cast<uint2*>(ranges), boundary_tile);
}

// exactly one of g_sph (expanded SH gradient) and g_radiance (view factor, see gut_project_bwd_kernel<true>) is non-null
void launch_project_bwd(hipStream_t s, const GutParams& P, const GutProjected& proj, const float* density12, const float* sph,
                        const float* g_rgb, const GutGradOut& g_out, float* g_sph, float* g_radiance, uint32_t first, uint32_t end) {
    if (end > P.N) end = P.N;
    if (end <= first) return;
    if (g_radiance)
        hipLaunchKernelGGL(gut_project_bwd_kernel<true>, dim3(div_up(end - first, 128)), dim3(128), 0, s, P, proj.tiles_count,
                           reinterpret_cast<const float4*>(density12), sph, proj.rgb, g_rgb, g_out, g_sph, g_radiance, first, end);
    else
        hipLaunchKernelGGL(gut_project_bwd_kernel<false>, dim3(div_up(end - first, 128)), dim3(128), 0, s, P, proj.tiles_count,
                           reinterpret_cast<const float4*>(density12), sph, proj.rgb, g_rgb, g_out, g_sph, g_radiance, first, end);
}
void launch_sph_grad_from_views(hipStream_t s, uint32_t N, uint32_t n_views, const float* factors, const float* positions, uint32_t pos_stride,
                                int n_active, int ncoef, float scale, float* g_sph) {
    hipLaunchKernelGGL(sph_grad_from_views_kernel, dim3(div_up(N, 128)), dim3(128), 0, s, N, n_views, factors, positions, pos_stride, n_active,
                       ncoef, scale, g_sph);
}
// neural harmonic features: the geometric gradient alone (the feature rows' gradient is accumulated by the sweep itself)
void launch_grad_finalize_nht(hipStream_t s, const GutParams& P, const GutProjected& proj, const float* density12, const GutGradSlots& slots,
                              bool have_partials, const GutGradOut& g_out) {
    hipLaunchKernelGGL((gut_grad_gather_kernel<16, true>), dim3(div_up(P.N, 64)), dim3(256), 0, s, P, proj, reinterpret_cast<const float4*>(density12),
                       slots, have_partials ? 1 : 0, g_out, static_cast<float*>(nullptr), 0u, P.N);
}
void launch_grad_finalize(hipStream_t s, const GutParams& P, const GutProjected& proj, const float* density12, const float* sph,
                          const GutGradSlots& slots, bool has_gdist, bool have_partials, float* g_rgb, const GutGradOut& g_out, float* g_sph,
                          float* g_radiance, uint32_t first, uint32_t end) {
    // particles [first, end) (first a multiple of 128): everything, or one chunk of the pipelined exchange
    if (end > P.N) end = P.N;
    if (end <= first) return;
    const dim3 grid(div_up(end - first, 64)), block(256);  // four lanes per particle
    static const bool unfused = [] { const char* e = getenv("GRUT_GUT_UNFUSED_FINALIZE"); return e && e[0] == '1'; }();   // (A/B measurements)
    if (!unfused && !g_radiance && g_sph && P.ncoef == 16 && !P.sph_half) {   // one kernel: gather + projection backward
        if (has_gdist)
            hipLaunchKernelGGL((gut_grad_gather_kernel<20, false, true>), grid, block, 0, s, P, proj, reinterpret_cast<const float4*>(density12), slots,
                               have_partials ? 1 : 0, g_out, g_rgb, first, end, sph, g_sph);
        else
            hipLaunchKernelGGL((gut_grad_gather_kernel<16, false, true>), grid, block, 0, s, P, proj, reinterpret_cast<const float4*>(density12), slots,
                               have_partials ? 1 : 0, g_out, g_rgb, first, end, sph, g_sph);
        return;
    }
    if (has_gdist)
        hipLaunchKernelGGL(gut_grad_gather_kernel<20>, grid, block, 0, s, P, proj, reinterpret_cast<const float4*>(density12), slots,
                           have_partials ? 1 : 0, g_out, g_rgb, first, end);
    else
        hipLaunchKernelGGL(gut_grad_gather_kernel<16>, grid, block, 0, s, P, proj, reinterpret_cast<const float4*>(density12), slots,
                           have_partials ? 1 : 0, g_out, g_rgb, first, end);
    launch_project_bwd(s, P, proj, density12, sph, g_rgb, g_out, g_sph, g_radiance, first, end);
}

}  // namespace grut
