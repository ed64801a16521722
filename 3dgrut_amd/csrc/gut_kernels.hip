// gut_kernels.hip — 3DGUT device code for gfx950: unscented projection onto tiles, ordered tile
// expansion, tile ranges, front-to-back compositing and its gradient sweep, projection backward.
//
// Reference behaviour restated (not translated): threedgut_tracer/include/3dgut/kernels/cuda/renderers/
// gutProjector.cuh:32-430, gutKBufferRenderer.cuh:199-352,642-716, common/rayPayload*.cuh,
// models/gaussianParticles.cuh:484-751.  CDNA4 design:
//   * binning key is (tile, depth-rank): particles are depth-sorted once (N keys), tile entries are
//     emitted in rank order, so only the tile bits need stable radix passes over the I entries;
//   * compositing runs one wave64 per 16x4 pixel strip (4 strips per 16x16 tile, same XCD), stages 64
//     tile entries per round in LDS with wave-synchronous hand-off (no workgroup barriers between
//     waves), and terminates per wave by ballot;
//   * the gradient sweep reduces each particle's 14 gradient terms over the wave with DPP adds and
//     flushes one atomic set per (strip, particle-with-hit) from an LDS accumulator.
#include <cstdlib>

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
__device__ __forceinline__ float tile_min_power(float tx, float ty, float4 co, float mx, float my) {
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
        const float rcpx = 1.f / (ts * ts * co.x), rcpy = 1.f / (ts * ts * co.z);
        const float tx_ = nry * saturate((dx * co.x * diffx + dx * co.y * diffy) * rcpx);
        const float ty_ = nrx * saturate((dy * co.y * diffx + dy * co.z * diffy) * rcpy);
        const float mdx = mx - (px + tx_ * dx), mdy = my - (py + ty_ * dy);
        return 0.5f * (co.x * mdx * mdx + co.z * mdy * mdy) + co.y * mdx * mdy;
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
__device__ __forceinline__ void coop_tile_walk(int lane, int gx, bool culling, TileBBox bb, float4 co, float cx, float cy, float pmax,
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

// ---------------------------------------------------------------------------------------------
// K1: projection onto tiles — GUTProjector::eval (gutProjector.cuh:217-322)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gut_project_kernel(GutParams P, const float4* __restrict__ density12,
                                                          const float* __restrict__ sph, GutProjected out,
                                                          int32_t* __restrict__ visibility, uint32_t* __restrict__ num_visible) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool has_tiles = false;
    uint32_t ntiles = 0;
    int vis = 0;
    float cx = 0.f, cy = 0.f, ex = 0.f, ey = 0.f, depth = 0.f, pmax_tile = 0.f, view_z_keep = 0.f;
    float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
    TileBBox bb = {0, 0, 0, 0};
    f3 pos = mk3(0.f, 0.f, 0.f);
    if (i < P.N) {
        const float4 a = density12[3 * (size_t)i + 0];  // pos.xyz, density
        const float4 b = density12[3 * (size_t)i + 1];  // quat wxyz
        const float4 c = density12[3 * (size_t)i + 2];  // scale.xyz, pad
        pos = mk3(a.x, a.y, a.z);
        const float opacity = a.w;

        const float view_z = fmaf(P.poses.view_R[6], pos.x, fmaf(P.poses.view_R[7], pos.y, fmaf(P.poses.view_R[8], pos.z, P.poses.view_t[2])));
        view_z_keep = view_z;
        bool ok = (opacity >= P.min_alpha) && (view_z >= 0.2f);
        if (ok) {
            const m3 rotT = quat_wxyz_to_rotT(b.x, b.y, b.z, b.w);
            float spx[7], spy[7];
            int nvalid = 0;
            nvalid += project_point_with_shutter(P.cam, P.poses, P.n_rs_iter, pos, P.ut_margin, spx[0], spy[0]) ? 1 : 0;
            cx = spx[0] * P.ut_w0m; cy = spy[0] * P.ut_w0m;
            const f3 axes[3] = {rotT.r0 * (P.ut_delta * c.x), rotT.r1 * (P.ut_delta * c.y), rotT.r2 * (P.ut_delta * c.z)};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                nvalid += project_point_with_shutter(P.cam, P.poses, P.n_rs_iter, pos + axes[k], P.ut_margin, spx[k + 1], spy[k + 1]) ? 1 : 0;
                cx += P.ut_wi * spx[k + 1]; cy += P.ut_wi * spy[k + 1];
                nvalid += project_point_with_shutter(P.cam, P.poses, P.n_rs_iter, pos - axes[k], P.ut_margin, spx[k + 4], spy[k + 4]) ? 1 : 0;
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
    // per-tile culling count (gutProjector.cuh:279-293), one particle at a time across the wave
    if (P.tile_culling) {
        unsigned long long todo = __ballot(vis != 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const TileBBox sb = {bcast_i(bb.minx, src), bcast_i(bb.miny, src), bcast_i(bb.maxx, src), bcast_i(bb.maxy, src)};
            const float4 sco = make_float4(bcast_f(co.x, src), bcast_f(co.y, src), bcast_f(co.z, src), 0.f);
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
        if (!has_tiles) {
            out.proj_pos[i] = make_float2(0.f, 0.f);
            out.conic_opacity[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            out.extent[i] = make_float2(0.f, 0.f);
            out.depth[i] = 0.f;
            out.depth_key[i] = 0xFFFFFFFFu;  // sorts behind every visible particle
        } else {
            const f3 ray = pos - mk3(P.poses.s2w_t[0], P.poses.s2w_t[1], P.poses.s2w_t[2]);
            const float dist = sqrtf(dot(ray, ray));
            const f3 dir = ray * (1.f / dist);
            float basis[16];
            sh_basis(P.n_active, dir, basis);
            const int nact = (P.n_active + 1) * (P.n_active + 1);
            const float* coef = sph + (size_t)i * 3 * P.ncoef;
            float r = 0.f, g = 0.f, bl = 0.f;
            for (int k = 0; k < nact && k < P.ncoef; ++k) {
                r = fmaf(basis[k], coef[3 * k + 0], r);
                g = fmaf(basis[k], coef[3 * k + 1], g);
                bl = fmaf(basis[k], coef[3 * k + 2], bl);
            }
            out.rgb[3 * (size_t)i + 0] = r + 0.5f;
            out.rgb[3 * (size_t)i + 1] = g + 0.5f;
            out.rgb[3 * (size_t)i + 2] = bl + 0.5f;
            out.proj_pos[i] = make_float2(cx, cy);
            out.conic_opacity[i] = co;
            out.extent[i] = make_float2(ex, ey);
            depth = P.global_z ? view_z_keep : dist;
            out.depth[i] = depth;
            out.depth_key[i] = __float_as_uint(depth);
        }
        out.particle_idx[i] = i;
    }
    // Nv for the byte model: one atomic per wave
    const unsigned long long m = __ballot(has_tiles);
    if (lane_id() == 0 && m) atomicAdd(num_visible, (uint32_t)__popcll(m));
}

// ---------------------------------------------------------------------------------------------
// K4: ordered expansion — GUTProjector::expand (gutProjector.cuh:324-388), iterated in depth-rank order
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gut_expand_kernel(GutParams P, GutProjected proj, const uint32_t* __restrict__ rank_to_particle,
                                                         const uint32_t* __restrict__ offsets, uint32_t capacity,
                                                         uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ tile_vals) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t off = 0, max_off = 0, p = 0;
    float cx = 0.f, cy = 0.f, pmax = 0.f;
    float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
    TileBBox bb = {0, 0, 0, 0};
    if (r < P.N) {
        off = r == 0 ? 0u : offsets[r - 1];
        max_off = min(offsets[r], capacity);
        if (max_off > off) {
            p = rank_to_particle[r];
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
    // one particle at a time across the wave: its entries land contiguously at [off, max_off) (coalesced stores);
    // their order inside the range is irrelevant, the tile sort that follows is stable per (tile) and each
    // (tile, particle) pair occurs once
    unsigned long long todo = __ballot(max_off > off);
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const TileBBox sb = {bcast_i(bb.minx, src), bcast_i(bb.miny, src), bcast_i(bb.maxx, src), bcast_i(bb.maxy, src)};
        const float4 sco = make_float4(bcast_f(co.x, src), bcast_f(co.y, src), bcast_f(co.z, src), 0.f);
        const uint32_t sp = (uint32_t)bcast_i((int)p, src), send = (uint32_t)bcast_i((int)max_off, src);
        uint32_t o = (uint32_t)bcast_i((int)off, src);
        coop_tile_walk(lane, P.gx, P.tile_culling != 0, sb, sco, bcast_f(cx, src), bcast_f(cy, src), bcast_f(pmax, src),
                       [&](bool keep, uint32_t tile) {
                           const unsigned long long m = __ballot(keep);
                           const uint32_t slot = o + (uint32_t)__popcll(m & lt_mask);
                           if (keep && slot < send) {
                               tile_keys[slot] = tile;
                               tile_vals[slot] = sp;
                           }
                           o += (uint32_t)__popcll(m);
                       });
        for (uint32_t k = o + lane; k < send; k += 64) {  // gutProjector.cuh:372-376 padding
            tile_keys[k] = 0xFFFFFFFFu;
            tile_vals[k] = 0xFFFFFFFFu;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K6: tile ranges — computeSortedTileRangeIndices (gutRenderer.cu:46-76)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gut_tile_ranges_kernel(uint32_t n, uint32_t tile_mask, uint32_t num_tiles,
                                                              const uint32_t* __restrict__ sorted_tile_keys, uint2* __restrict__ ranges,
                                                              uint32_t* __restrict__ boundary_tile) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t t = sorted_tile_keys[k] & tile_mask;
    const bool valid = t < num_tiles;
    // segment boundary b sits at sorted index b * kGutSegment; remember which tile's list it cuts
    if ((k % kGutSegment) == 0) boundary_tile[k / kGutSegment] = valid ? t : 0xFFFFFFFFu;
    if (k == 0) {
        if (valid) ranges[t].x = 0;
    } else {
        const uint32_t pt = sorted_tile_keys[k - 1] & tile_mask;
        if (pt != t) {
            if (pt < num_tiles) ranges[pt].y = k;
            if (valid) ranges[t].x = k;
        }
    }
    if (valid && k == n - 1) ranges[t].y = n;
}

// ---------------------------------------------------------------------------------------------
// rays (rayPayload.cuh:75-108, bounding_box.h:89-140)
// ---------------------------------------------------------------------------------------------
struct Ray {
    f3 o, d;
    float tmin, tmax;
    bool valid;
};
__device__ __forceinline__ void swapf(float& a, float& b) { const float t = a; a = b; b = t; }
__device__ __forceinline__ Ray init_ray(const GutParams& P, const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                        int px, int py) {
    Ray r;
    r.valid = false;
    r.o = r.d = mk3(0.f, 0.f, 0.f);
    r.tmin = r.tmax = 0.f;
    if (px >= P.W || py >= P.H) return r;
    const size_t pix = (size_t)py * P.W + px;
    const f3 so = mk3(ray_o[3 * pix], ray_o[3 * pix + 1], ray_o[3 * pix + 2]);
    const f3 sd = mk3(ray_d[3 * pix], ray_d[3 * pix + 1], ray_d[3 * pix + 2]);
    const float* R = P.poses.s2w_R;
    r.o = apply_rows(R, P.poses.s2w_t, so);
    r.d = mk3(fmaf(R[0], sd.x, fmaf(R[1], sd.y, R[2] * sd.z)), fmaf(R[3], sd.x, fmaf(R[4], sd.y, R[5] * sd.z)),
              fmaf(R[6], sd.x, fmaf(R[7], sd.y, R[8] * sd.z)));
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

// strip -> (tile, strip-in-tile) with all four strips of a tile on one XCD (block b runs on XCD b % 8)
__device__ __forceinline__ bool strip_mapping(uint32_t b, uint32_t num_tiles, uint32_t& tile, uint32_t& strip) {
    const uint32_t xcd = b & 7u, slot = b >> 3;
    tile = ((slot >> 2) << 3) + xcd;
    strip = slot & 3u;
    return tile < num_tiles;
}


// ---------------------------------------------------------------------------------------------
// K7: compositing forward — GUTKBufferRenderer::evalKBuffer, K = 0 (gutKBufferRenderer.cuh:273-352)
// one wave64 = 16x4 pixels; LDS record per staged tile entry: 5 x float4
//   q0 = M.r0, pos.x | q1 = M.r1, pos.y | q2 = M.r2, pos.z | q3 = scale.xyz, density | q4 = rgb(clamped), -
//   with M = diag(1/scale) * R^T  (canonical-space transform, gaussianParticles.slang:96-110)
// Staging rounds are aligned to multiples of 64 in the global sorted list, so every segment boundary
// (multiple of kGutSegment) is a round start, where the running state is checkpointed for the gradient sweep.
// ---------------------------------------------------------------------------------------------
template <int DEG, bool CKPT>
__global__ __launch_bounds__(64) void gut_render_fwd_kernel(GutParams P, const uint2* __restrict__ ranges,
                                                            const uint32_t* __restrict__ sorted_idx,
                                                            const float4* __restrict__ density12, const float* __restrict__ rgb,
                                                            const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                            float4* __restrict__ out_fd, float* __restrict__ out_dist,
                                                            float* __restrict__ out_cnt, GutCheckpoints ck) {
    __shared__ float4 s_rec[64 * 5];
    uint32_t tile, strip;
    if (!strip_mapping(blockIdx.x, P.gx * P.gy, tile, strip)) return;
    const int lane = threadIdx.x;
    const int px = (int)(tile % P.gx) * 16 + (lane & 15);
    const int py = (int)(tile / P.gx) * 16 + (int)strip * 4 + (lane >> 4);
    const Ray ray = init_ray(P, ray_o, ray_d, px, py);
    bool alive = ray.valid;

    const uint2 range = ranges[tile];
    float T = 1.f, D = 0.f, Cr = 0.f, Cg = 0.f, Cb = 0.f;
    uint32_t cnt = 0;

    uint32_t b = range.x;
    while (b < range.y) {
        if (!__any(alive)) break;
        const uint32_t bend = min(range.y, (b & ~63u) + 64u);
        if (CKPT && b > range.x && (b % kGutSegment) == 0) {
            const size_t slot = ((size_t)(b / kGutSegment) * 4 + strip) * 64 + lane;
            ck.tc[slot] = make_float4(alive ? T : 0.f, Cr, Cg, Cb);  // dead lanes restart dead (T = 0 < min_transmittance)
            ck.d[slot] = D;
            if (lane == 0) ck.reached[(size_t)(b / kGutSegment) * 4 + strip] = 1;
        }
        {   // stage up to 64 entries
            const uint32_t e = b + lane;
            float4 q0, q1, q2, q3, q4;
            q0 = q1 = q2 = make_float4(0.f, 0.f, 0.f, 0.f);
            q3 = make_float4(1.f, 1.f, 1.f, 0.f);
            q4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < bend) {
                const uint32_t idx = sorted_idx[e];
                if (idx != 0xFFFFFFFFu) {
                    const float4 a = density12[3 * (size_t)idx + 0];
                    const float4 q = density12[3 * (size_t)idx + 1];
                    const float4 s = density12[3 * (size_t)idx + 2];
                    const m3 rt = quat_wxyz_to_rotT(q.x, q.y, q.z, q.w);
                    const float ix = __builtin_amdgcn_rcpf(s.x), iy = __builtin_amdgcn_rcpf(s.y), iz = __builtin_amdgcn_rcpf(s.z);
                    q0 = make_float4(rt.r0.x * ix, rt.r0.y * ix, rt.r0.z * ix, a.x);
                    q1 = make_float4(rt.r1.x * iy, rt.r1.y * iy, rt.r1.z * iy, a.y);
                    q2 = make_float4(rt.r2.x * iz, rt.r2.y * iz, rt.r2.z * iz, a.z);
                    q3 = make_float4(s.x, s.y, s.z, a.w);
                    q4 = make_float4(fmaxf(rgb[3 * (size_t)idx], 0.f), fmaxf(rgb[3 * (size_t)idx + 1], 0.f),
                                     fmaxf(rgb[3 * (size_t)idx + 2], 0.f), 0.f);
                }
            }
            s_rec[lane * 5 + 0] = q0; s_rec[lane * 5 + 1] = q1; s_rec[lane * 5 + 2] = q2;
            s_rec[lane * 5 + 3] = q3; s_rec[lane * 5 + 4] = q4;
        }
        __syncthreads();  // single-wave workgroup: orders the LDS hand-off
        const int n = (int)(bend - b);
        for (int j = 0; j < n; ++j) {
            if (!alive) continue;
            const float4 q0 = s_rec[j * 5 + 0], q1 = s_rec[j * 5 + 1], q2 = s_rec[j * 5 + 2], q3 = s_rec[j * 5 + 3];
            const f3 dlt = ray.o - mk3(q0.w, q1.w, q2.w);
            const f3 gro = mk3(dot(mk3(q0.x, q0.y, q0.z), dlt), dot(mk3(q1.x, q1.y, q1.z), dlt), dot(mk3(q2.x, q2.y, q2.z), dlt));
            const f3 grdu = mk3(dot(mk3(q0.x, q0.y, q0.z), ray.d), dot(mk3(q1.x, q1.y, q1.z), ray.d), dot(mk3(q2.x, q2.y, q2.z), ray.d));
            const f3 grd = grdu * __builtin_amdgcn_rsqf(dot(grdu, grdu));
            const f3 gc = cross(grd, gro);
            const float gray = dot(gc, gc);
            const float resp = particle_response<DEG>(gray);
            const float alpha = fminf(P.max_alpha, resp * q3.w);
            if ((resp > P.min_response) && (alpha > P.min_alpha)) {
                const float pd = -dot(grd, gro);
                const f3 grds = mk3(q3.x, q3.y, q3.z) * grd * pd;
                const float hitT = __builtin_amdgcn_sqrtf(dot(grds, grds));
                if ((hitT > ray.tmin) && (hitT < ray.tmax)) {
                    const float4 q4 = s_rec[j * 5 + 4];
                    const float w = alpha * T;
                    D = fmaf(hitT, w, D);
                    T *= (1.f - alpha);
                    if (w > 0.f) {
                        Cr = fmaf(q4.x, w, Cr); Cg = fmaf(q4.y, w, Cg); Cb = fmaf(q4.z, w, Cb);
                        cnt++;
                    }
                    if (T < P.min_transmittance) alive = false;
                }
            }
        }
        __syncthreads();
        b = bend;
    }
    if (ray.valid) {
        const size_t pix = (size_t)py * P.W + px;
        out_fd[pix] = make_float4(Cr, Cg, Cb, 1.f - T);
        out_dist[pix] = D;
        if (P.hitcounts) out_cnt[pix] = (float)cnt;
    }
}

// ---------------------------------------------------------------------------------------------
// K8: compositing backward — evalBackwardNoKBuffer SH branch (gutKBufferRenderer.cuh:642-716) with
// processHitBwd (models/gaussianParticles.cuh:484-751).  LDS record: 7 x float4
//   q0 = rotT.r0, pos.x | q1 = rotT.r1, pos.y | q2 = rotT.r2, pos.z | q3 = scale.xyz, density
//   q4 = 2 * quat wxyz  | q5 = rgb(clamped), as_float(idx) | q6 = 1/scale.xyz, -
//
// Per hit every lane produces 14 terms that are summed over the wave (wave_reduce_scatter16) and flushed with
// one atomic set per (strip, particle-with-hit):
//   B[3]   = d L / d (R^T (o - mu))            -> position gradient = -R B      (applied once, at flush)
//   dn     = d L / d density
//   dq[4]  = d L / d quaternion (w,x,y,z)
//   S[3]   with scale gradient = -S / scale                                     (applied once, at flush)
//   dc[3]  = d L / d (clamped particle radiance)
//
// With u = gro, v = grdu, n = v/|v|:  grayDist = |n x u|^2 = |u|^2 - (n.u)^2, hence
//   d gray / d u = 2 a,   d gray / d v = 2 beta a,   a = u - (n.u) n,   beta = -(n.u)/|v|
// i.e. every geometric gradient that flows through grayDist is a multiple of the single vector a.  The reference's
// chain (two cross-product backward passes, safe_normalize_bw, two matmul_bw_quat) collapses to one rank-1
// contraction  d rotT = (giscl*a) (x) (gposc + beta d)  — same mathematics, ~4x fewer instructions.  When a depth
// gradient flows in (HAS_GDIST) the hit-distance terms are added in their generic form.
// ---------------------------------------------------------------------------------------------
// gradient of sum_ij b_i e_j rotT_ij(q) w.r.t. q = (r,x,y,z); q2 = 2q (matmul_bw_quat, mathUtils.cuh:458-521)
__device__ __forceinline__ void quat_contract(f3 b, f3 e, float4 q2, float& dr, float& dx, float& dy, float& dz) {
    const float r = q2.x, x = q2.y, y = q2.z, z = q2.w;
    const float m00 = b.x * e.x, m01 = b.x * e.y, m02 = b.x * e.z;
    const float m10 = b.y * e.x, m11 = b.y * e.y, m12 = b.y * e.z;
    const float m20 = b.z * e.x, m21 = b.z * e.y, m22 = b.z * e.z;
    // rotT = [[1-2(yy+zz), 2(xy+rz), 2(xz-ry)], [2(xy-rz), 1-2(xx+zz), 2(yz+rx)], [2(xz+ry), 2(yz-rx), 1-2(xx+yy)]]
    const float s01 = m01 + m10, s02 = m02 + m20, s12 = m12 + m21;   // symmetric parts
    const float a01 = m01 - m10, a02 = m20 - m02, a12 = m12 - m21;   // antisymmetric parts (signs as in rotT)
    dr += z * a01 + y * a02 + x * a12;
    dx += y * s01 + z * s02 + r * a12 - 2.f * x * (m11 + m22);
    dy += x * s01 + z * s12 + r * a02 - 2.f * y * (m00 + m22);
    dz += x * s02 + y * s12 + r * a01 - 2.f * z * (m00 + m11);
}

template <int DEG, bool HAS_GDIST, bool COUNT = false>
__global__ __launch_bounds__(64) void gut_render_bwd_kernel(GutParams P, const uint2* __restrict__ ranges,
                                                            const uint32_t* __restrict__ sorted_idx,
                                                            const float4* __restrict__ density12, const float* __restrict__ rgb,
                                                            const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                            const float4* __restrict__ fd, const float4* __restrict__ g_fd,
                                                            const float* __restrict__ dist, const float* __restrict__ g_dist,
                                                            float* __restrict__ g_density12, float* __restrict__ g_rgb,
                                                            GutCheckpoints ck, unsigned long long* __restrict__ counters = nullptr) {
    // 32 staged entries per round: 5.5 KB of LDS per wave keeps ~7 waves per SIMD resident (the sweep is
    // latency-bound at 3-4 waves per SIMD: rocprofv3 showed 45% SQ_WAIT_INST_ANY with 64-entry rounds)
    constexpr uint32_t kBatch = 32;
    __shared__ float4 s_rec[kBatch * 7];
    __shared__ float s_acc[kBatch * 16];  // per staged entry: 14 wave-reduced terms (+2 pad)
    // task = (virtual tile, strip): virtual tiles [0, tilesPad) are the first segment of each tile list, virtual tile
    // tilesPad + b is the segment starting at segment boundary b (sorted index b * kGutSegment)
    const uint32_t num_tiles = (uint32_t)(P.gx * P.gy), tiles_pad = (num_tiles + 7u) & ~7u;
    uint32_t vtile, strip;
    (void)strip_mapping(blockIdx.x, 0xFFFFFFFFu, vtile, strip);
    uint32_t tile, seg_begin;
    bool from_checkpoint = false;
    uint32_t boundary = 0;
    if (vtile < tiles_pad) {
        tile = vtile;
        if (tile >= num_tiles) return;
        seg_begin = ranges[tile].x;
    } else {
        boundary = vtile - tiles_pad;
        if (boundary == 0 || boundary >= ck.num_boundaries) return;
        tile = ck.boundary_tile[boundary];
        if (tile >= num_tiles) return;
        seg_begin = boundary * kGutSegment;
        if (seg_begin <= ranges[tile].x) return;                     // the boundary is this tile's own list start
        if (!ck.reached[(size_t)boundary * 4 + strip]) return;       // the forward sweep never got here alive
        from_checkpoint = true;
    }
    const uint32_t seg_end = min(ranges[tile].y, (seg_begin / kGutSegment + 1u) * kGutSegment);
    const int lane = threadIdx.x;
    const int px = (int)(tile % P.gx) * 16 + (lane & 15);
    const int py = (int)(tile / P.gx) * 16 + (int)strip * 4 + (lane >> 4);
    const Ray ray = init_ray(P, ray_o, ray_d, px, py);
    bool alive = ray.valid;

    float T = 1.f, D = 0.f, Cr = 0.f, Cg = 0.f, Cb = 0.f;
    float T_fin = 0.f, D_fin = 0.f, gT = 0.f, gD = 0.f;
    f3 C_fin = mk3(0.f, 0.f, 0.f), gC = mk3(0.f, 0.f, 0.f);
    if (alive) {
        const size_t pix = (size_t)py * P.W + px;
        const float4 f = fd[pix], g = g_fd[pix];
        C_fin = mk3(f.x, f.y, f.z); gC = mk3(g.x, g.y, g.z);
        T_fin = 1.f - f.w; gT = -g.w;
        if (HAS_GDIST) { D_fin = dist[pix]; gD = g_dist[pix]; }
    }
    if (from_checkpoint) {
        const size_t slot = ((size_t)boundary * 4 + strip) * 64 + lane;
        const float4 c = ck.tc[slot];
        T = c.x; Cr = c.y; Cg = c.z; Cb = c.w;
        if (HAS_GDIST) D = ck.d[slot];
        alive = alive && !(T < P.min_transmittance);
    }

    uint32_t b = seg_begin;
    while (b < seg_end) {
        if (!__any(alive)) break;
        const uint32_t bend = min(seg_end, (b & ~(kBatch - 1u)) + kBatch);
        if (lane < (int)kBatch) {
            const uint32_t e = b + lane;
            float4 q0, q1, q2, q3, q4, q5, q6;
            q0 = q1 = q2 = q4 = make_float4(0.f, 0.f, 0.f, 0.f);
            q3 = make_float4(1.f, 1.f, 1.f, 0.f);
            q5 = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
            q6 = make_float4(1.f, 1.f, 1.f, 0.f);
            if (e < bend) {
                const uint32_t idx = sorted_idx[e];
                if (idx != 0xFFFFFFFFu) {
                    const float4 a = density12[3 * (size_t)idx + 0];
                    const float4 q = density12[3 * (size_t)idx + 1];
                    const float4 s = density12[3 * (size_t)idx + 2];
                    const m3 rt = quat_wxyz_to_rotT(q.x, q.y, q.z, q.w);
                    q0 = make_float4(rt.r0.x, rt.r0.y, rt.r0.z, a.x);
                    q1 = make_float4(rt.r1.x, rt.r1.y, rt.r1.z, a.y);
                    q2 = make_float4(rt.r2.x, rt.r2.y, rt.r2.z, a.z);
                    q3 = make_float4(s.x, s.y, s.z, a.w);
                    q4 = make_float4(2.f * q.x, 2.f * q.y, 2.f * q.z, 2.f * q.w);
                    q5 = make_float4(fmaxf(rgb[3 * (size_t)idx], 0.f), fmaxf(rgb[3 * (size_t)idx + 1], 0.f),
                                     fmaxf(rgb[3 * (size_t)idx + 2], 0.f), __uint_as_float(idx));
                    q6 = make_float4(__builtin_amdgcn_rcpf(s.x), __builtin_amdgcn_rcpf(s.y), __builtin_amdgcn_rcpf(s.z), 0.f);
                }
            }
            float4* r = &s_rec[lane * 7];
            r[0] = q0; r[1] = q1; r[2] = q2; r[3] = q3; r[4] = q4; r[5] = q5; r[6] = q6;
        }
        __syncthreads();
        const int n = (int)(bend - b);
        uint32_t hit_entries = 0u;  // wave-uniform: staged entries with >= 1 hit in this wave
        for (int j = 0; j < n; ++j) {
            if (!__any(alive)) break;
            float t_bx = 0.f, t_by = 0.f, t_bz = 0.f, t_dn = 0.f, t_qr = 0.f, t_qx = 0.f, t_qy = 0.f, t_qz = 0.f;
            float t_sx = 0.f, t_sy = 0.f, t_sz = 0.f, t_cr = 0.f, t_cg = 0.f, t_cb = 0.f;
            bool hit = false;
            if (alive) {
                const float4* rec = &s_rec[j * 7];
                const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q6 = rec[6];
                const m3 rotT = {mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q2.x, q2.y, q2.z)};
                const f3 gscl = mk3(q3.x, q3.y, q3.z), giscl = mk3(q6.x, q6.y, q6.z);
                const float dens = q3.w;
                const f3 gposc = ray.o - mk3(q0.w, q1.w, q2.w);
                const f3 u = giscl * mul_rows(rotT, gposc);           // gro
                const f3 v = giscl * mul_rows(rotT, ray.d);           // grdu
                const float l2 = dot(v, v);
                const float il = l2 > 0.f ? __builtin_amdgcn_rsqf(l2) : 1.f;  // safe_normalize
                const f3 nrm = v * il;                                 // grd
                const float nu = dot(nrm, u);
                const f3 avec = u - nrm * nu;                          // component of u orthogonal to the ray
                const float gray = dot(avec, avec);                    // == |grd x gro|^2
                const float gres = particle_response<DEG>(gray);
                const float galpha = fminf(P.max_alpha, gres * dens);
                if ((gres > P.min_response) && (galpha > P.min_alpha)) {
                    hit = true;
                    const float4 q4 = rec[4], q5 = rec[5];
                    const f3 feat = mk3(q5.x, q5.y, q5.z);
                    const float weight = galpha * T;
                    const float nextT = (1.f - galpha) * T;
                    const float inextT = nextT <= P.min_transmittance ? 0.f : __builtin_amdgcn_rcpf(nextT);
                    const float resTrm = galpha < 0.999999f ? T_fin * __builtin_amdgcn_rcpf(1.f - galpha) : T;
                    float dalpha = resTrm * -gT;  // d L / d alpha
                    t_cr = gC.x * weight; t_cg = gC.y * weight; t_cb = gC.z * weight;
                    Cr = fmaf(feat.x, weight, Cr); Cg = fmaf(feat.y, weight, Cg); Cb = fmaf(feat.z, weight, Cb);
                    const f3 resRad = mk3(fmaxf((C_fin.x - Cr) * inextT, 0.f), fmaxf((C_fin.y - Cg) * inextT, 0.f),
                                          fmaxf((C_fin.z - Cb) * inextT, 0.f));
                    dalpha += T * ((feat.x - resRad.x) * gC.x + (feat.y - resRad.y) * gC.y + (feat.z - resRad.z) * gC.z);

                    // hit-distance terms (models/gaussianParticles.cuh:545-580), generic form
                    f3 u_extra = mk3(0.f, 0.f, 0.f), v_extra = mk3(0.f, 0.f, 0.f), s_extra = mk3(0.f, 0.f, 0.f);
                    if (HAS_GDIST) {
                        const float pdot = -nu;
                        const f3 grdd = nrm * pdot;
                        const f3 grds = gscl * grdd;
                        const float gsq = dot(grds, grds);
                        const float gdist = __builtin_amdgcn_sqrtf(gsq);
                        D = fmaf(weight, gdist, D);
                        const float resHitT = fmaxf((D_fin - D) * inextT, 0.f);
                        dalpha += (gdist - resHitT) * T * gD;
                        const f3 grdsGrd = gsq > 0.f ? grds * (weight * __builtin_amdgcn_rcpf(gdist) * gD) : mk3(0.f, 0.f, 0.f);
                        s_extra = grdd * grdsGrd;                              // direct d hitT / d scale
                        const float sd = dot(grdsGrd * gscl, nrm);
                        const f3 nGrd = (gscl * grdsGrd) * pdot - u * sd;       // d / d grd
                        u_extra = nrm * (-sd);                                   // d / d gro
                        // safe_normalize_bw: (g - n (n.g)) / |v|
                        v_extra = l2 > 0.f ? (nGrd - nrm * dot(nrm, nGrd)) * il : mk3(0.f, 0.f, 0.f);
                    }

                    t_dn = gres * dalpha;
                    const float w2 = 2.f * particle_response_grd<DEG>(gray, gres, dens * dalpha);  // 2 dL/dgray
                    const float beta = l2 > 0.f ? -nu * il : 0.f;
                    f3 uGrd = avec * w2;             // d L / d gro
                    f3 vGrd = uGrd * beta;           // d L / d grdu
                    if (HAS_GDIST) { uGrd = uGrd + u_extra; vGrd = vGrd + v_extra; }
                    const f3 bu = giscl * uGrd, bv = giscl * vGrd;   // d L / d (R^T gposc), d L / d (R^T d)
                    t_bx = bu.x; t_by = bu.y; t_bz = bu.z;
                    // scale: gro_i, grdu_i ~ 1/s_i  ->  d/ds_i = -(gro_i uGrd_i + grdu_i vGrd_i)/s_i (+ direct hitT term)
                    t_sx = fmaf(u.x, uGrd.x, v.x * vGrd.x); t_sy = fmaf(u.y, uGrd.y, v.y * vGrd.y); t_sz = fmaf(u.z, uGrd.z, v.z * vGrd.z);
                    if (HAS_GDIST) { t_sx -= gscl.x * s_extra.x; t_sy -= gscl.y * s_extra.y; t_sz -= gscl.z * s_extra.z; }
                    if (HAS_GDIST) {
                        quat_contract(bu, gposc, q4, t_qr, t_qx, t_qy, t_qz);
                        quat_contract(bv, ray.d, q4, t_qr, t_qx, t_qy, t_qz);
                    } else {  // bv = beta * bu: rank-1
                        quat_contract(bu, gposc + ray.d * beta, q4, t_qr, t_qx, t_qy, t_qz);
                    }
                    T = nextT;
                    if (T < P.min_transmittance) alive = false;
                }
            }
            if (COUNT) {
                const unsigned long long hm = __ballot(hit), am = __ballot(alive);
                if (lane == 0) {
                    atomicAdd(&counters[0], 1ull);                              // (strip, entry) pairs processed
                    if (hm) atomicAdd(&counters[1], 1ull);                      // ... with at least one hit
                    atomicAdd(&counters[2], (unsigned long long)__popcll(hm));  // hits
                    atomicAdd(&counters[3], (unsigned long long)__popcll(am));  // lanes still alive after the entry
                }
            }
            if (__any(hit)) {
                hit_entries |= (1u << j);
                const float terms[16] = {t_bx, t_by, t_bz, t_dn, t_qr, t_qx, t_qy, t_qz, t_sx, t_sy, t_sz, 0.f, t_cr, t_cg, t_cb, 0.f};
                const float tot = wave_reduce_scatter16(terms, lane);
                if (lane < 16) s_acc[j * 16 + lane] = tot;
            }
        }
        __syncthreads();
        // flush: lane j owns staged entry j; one atomic set per (strip, particle with a hit)
        if (lane < (int)kBatch && ((hit_entries >> lane) & 1u)) {
            const float4* rec = &s_rec[lane * 7];
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q6 = rec[6];
            const uint32_t idx = __float_as_uint(rec[5].w);
            const float4* acc = reinterpret_cast<const float4*>(&s_acc[lane * 16]);
            const float4 a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
            // position = -R B  (matmul_bw_vec with rows of R^T), scale = -S / scale
            const float gpx = -(a0.x * q0.x + a0.y * q1.x + a0.z * q2.x);
            const float gpy = -(a0.x * q0.y + a0.y * q1.y + a0.z * q2.y);
            const float gpz = -(a0.x * q0.z + a0.y * q1.z + a0.z * q2.z);
            float* gd = g_density12 + 12 * (size_t)idx;
            atomicAdd(gd + 0, gpx); atomicAdd(gd + 1, gpy); atomicAdd(gd + 2, gpz); atomicAdd(gd + 3, a0.w);
            atomicAdd(gd + 4, a1.x); atomicAdd(gd + 5, a1.y); atomicAdd(gd + 6, a1.z); atomicAdd(gd + 7, a1.w);
            atomicAdd(gd + 8, -a2.x * q6.x); atomicAdd(gd + 9, -a2.y * q6.y); atomicAdd(gd + 10, -a2.z * q6.z);
            float* gr = g_rgb + 3 * (size_t)idx;
            atomicAdd(gr + 0, a3.x); atomicAdd(gr + 1, a3.y); atomicAdd(gr + 2, a3.z);
        }
        __syncthreads();
        b = bend;
    }
}

// ---------------------------------------------------------------------------------------------
// K9: projection backward — GUTProjector::evalBackward (gutProjector.cuh:390-430): per visible particle
// dRGB -> dSH (clamp-masked) and d direction -> d position; every SH gradient row is written exactly once
// (zeros for particles without tiles), so the caller does not need to zero-fill grad_sph.
// ---------------------------------------------------------------------------------------------
// SH rows travel through LDS (row stride 49 floats: conflict-free both ways): a wave reads the coefficient rows of
// its 64 particles and writes their gradient rows as contiguous 192-byte segments instead of 64 scattered rows.
constexpr int kShStride = 49;
__global__ __launch_bounds__(128) void gut_project_bwd_kernel(GutParams P, const uint32_t* __restrict__ tiles_count,
                                                              const float4* __restrict__ density12, const float* __restrict__ sph,
                                                              const float* __restrict__ rgb, const float* __restrict__ g_rgb,
                                                              float* __restrict__ g_density12, float* __restrict__ g_sph) {
    __shared__ float s_rows[2][64 * kShStride];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t wave_base = blockIdx.x * 128u + (uint32_t)wave * 64u;
    const uint32_t i = wave_base + lane;
    const int rowlen = 3 * P.ncoef;
    const int nact = min((P.n_active + 1) * (P.n_active + 1), P.ncoef);
    float* rows = s_rows[wave];
    const bool has = (i < P.N) && (tiles_count[i] != 0);
    const unsigned long long has_mask = __ballot(has);
    // stage in: one row per step, lanes across the row
    for (unsigned long long m = has_mask; m;) {
        const int row = __ffsll((long long)m) - 1;
        m &= m - 1;
        if (lane < 3 * nact) rows[row * kShStride + lane] = sph[(size_t)(wave_base + row) * rowlen + lane];
    }
    __syncthreads();
    float* myrow = rows + lane * kShStride;
    if (has) {
        const float4 a = density12[3 * (size_t)i];
        const f3 v = mk3(a.x, a.y, a.z) - mk3(P.poses.s2w_t[0], P.poses.s2w_t[1], P.poses.s2w_t[2]);
        const float len = sqrtf(dot(v, v));
        const float ilen = 1.f / len;
        const f3 dir = v * ilen;
        f3 g = mk3(g_rgb[3 * (size_t)i], g_rgb[3 * (size_t)i + 1], g_rgb[3 * (size_t)i + 2]);
        // clamp mask on the unclamped radiance stored by the forward projection
        if (!(rgb[3 * (size_t)i] > 0.f)) g.x = 0.f;
        if (!(rgb[3 * (size_t)i + 1] > 0.f)) g.y = 0.f;
        if (!(rgb[3 * (size_t)i + 2] > 0.f)) g.z = 0.f;
        float basis[16];
        f3 dbasis[16];
        sh_basis(P.n_active, dir, basis);
        sh_basis_grad(P.n_active, dir, dbasis);
        f3 gdir = mk3(0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < nact) {
                const float s = g.x * myrow[3 * k] + g.y * myrow[3 * k + 1] + g.z * myrow[3 * k + 2];
                gdir = gdir + dbasis[k] * s;
                myrow[3 * k] = basis[k] * g.x; myrow[3 * k + 1] = basis[k] * g.y; myrow[3 * k + 2] = basis[k] * g.z;
            }
        }
        for (int k = 3 * nact; k < rowlen; ++k) myrow[k] = 0.f;
        const float ng = dot(dir, gdir);
        const f3 gpos = (gdir - dir * ng) * ilen;
        g_density12[12 * (size_t)i + 0] += gpos.x;
        g_density12[12 * (size_t)i + 1] += gpos.y;
        g_density12[12 * (size_t)i + 2] += gpos.z;
    } else {
        for (int k = 0; k < rowlen; ++k) myrow[k] = 0.f;
    }
    __syncthreads();
    // stage out: every row of the wave, zeros included (the caller does not pre-fill grad_sph)
    const int nrows = (int)min(64u, P.N > wave_base ? P.N - wave_base : 0u);
    for (int row = 0; row < nrows; ++row)
        if (lane < rowlen) g_sph[(size_t)(wave_base + row) * rowlen + lane] = rows[row * kShStride + lane];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void launch_project(hipStream_t s, const GutParams& P, const float* density12, const float* sph, const GutProjected& out,
                    int32_t* visibility, uint32_t* num_visible) {
    hipLaunchKernelGGL(gut_project_kernel, dim3(div_up(P.N, 256)), dim3(256), 0, s, P, reinterpret_cast<const float4*>(density12), sph,
                       out, visibility, num_visible);
}
void launch_expand(hipStream_t s, const GutParams& P, const GutProjected& proj, const uint32_t* rank_to_particle,
                   const uint32_t* offsets, uint32_t capacity, uint32_t* tile_keys, uint32_t* tile_vals) {
    hipLaunchKernelGGL(gut_expand_kernel, dim3(div_up(P.N, 256)), dim3(256), 0, s, P, proj, rank_to_particle, offsets, capacity,
                       tile_keys, tile_vals);
}
void launch_tile_ranges(hipStream_t s, uint32_t n, uint32_t tile_mask, uint32_t num_tiles, const uint32_t* sorted_tile_keys,
                        uint32_t* ranges, uint32_t* boundary_tile) {
    hipLaunchKernelGGL(gut_tile_ranges_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, tile_mask, num_tiles, sorted_tile_keys,
                       reinterpret_cast<uint2*>(ranges), boundary_tile);
}

static uint32_t strip_grid(const GutParams& P) {
    const uint32_t tiles = (uint32_t)(P.gx * P.gy);
    return ((tiles + 7u) & ~7u) * 4u;
}
// tile-first segments + one task set per segment boundary, padded to the 8-XCD interleave
static uint32_t segment_grid(const GutParams& P, uint32_t num_boundaries) {
    const uint32_t tiles = (uint32_t)(P.gx * P.gy);
    const uint32_t vtiles = ((tiles + 7u) & ~7u) + ((num_boundaries + 7u) & ~7u);
    return vtiles * 4u;
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

void launch_render_fwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_idx, const float* density12,
                       const float* rgb, const float* ray_o, const float* ray_d, float* out_fd, float* out_dist, float* out_cnt,
                       const GutCheckpoints& ck, bool write_checkpoints) {
    if (write_checkpoints) {
        GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_fwd_kernel<D_, true>), dim3(strip_grid(P)), dim3(64), 0, s, P,
                                                          reinterpret_cast<const uint2*>(ranges), sorted_idx,
                                                          reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d,
                                                          reinterpret_cast<float4*>(out_fd), out_dist, out_cnt, ck));
    } else {
        GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_fwd_kernel<D_, false>), dim3(strip_grid(P)), dim3(64), 0, s, P,
                                                          reinterpret_cast<const uint2*>(ranges), sorted_idx,
                                                          reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d,
                                                          reinterpret_cast<float4*>(out_fd), out_dist, out_cnt, ck));
    }
}
void launch_render_bwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_idx, const float* density12,
                       const float* rgb, const float* ray_o, const float* ray_d, const float* fd, const float* g_fd, const float* dist,
                       const float* g_dist, float* g_density12, float* g_rgb, const GutCheckpoints& ck) {
    const dim3 grid(segment_grid(P, ck.num_boundaries));
    if (getenv("GRUT_COUNT_HITS") && P.degree == 2) {  // development aid: work statistics of the gradient sweep
        unsigned long long* d = nullptr;
        unsigned long long hcnt[4] = {0, 0, 0, 0};
        (void)hipMalloc(&d, 32);
        (void)hipMemsetAsync(d, 0, 32, s);
        hipLaunchKernelGGL((gut_render_bwd_kernel<2, false, true>), grid, dim3(64), 0, s, P, reinterpret_cast<const uint2*>(ranges),
                           sorted_idx, reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d, reinterpret_cast<const float4*>(fd),
                           reinterpret_cast<const float4*>(g_fd), dist, g_dist, g_density12, g_rgb, ck, d);
        (void)hipMemcpyAsync(hcnt, d, 32, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        (void)hipFree(d);
        fprintf(stderr, "[grut] bwd strip-entries processed %llu, with>=1 hit %llu, hits %llu, alive-lane-entries %llu\n", hcnt[0], hcnt[1],
                hcnt[2], hcnt[3]);
        return;
    }
    if (g_dist) {
        GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_bwd_kernel<D_, true>), grid, dim3(64), 0, s, P,
                                                          reinterpret_cast<const uint2*>(ranges), sorted_idx,
                                                          reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d,
                                                          reinterpret_cast<const float4*>(fd), reinterpret_cast<const float4*>(g_fd), dist,
                                                          g_dist, g_density12, g_rgb, ck, nullptr));
    } else {  // no depth gradient flows in: the hit-distance terms vanish identically
        GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((gut_render_bwd_kernel<D_, false>), grid, dim3(64), 0, s, P,
                                                          reinterpret_cast<const uint2*>(ranges), sorted_idx,
                                                          reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d,
                                                          reinterpret_cast<const float4*>(fd), reinterpret_cast<const float4*>(g_fd), dist,
                                                          g_dist, g_density12, g_rgb, ck, nullptr));
    }
}
void launch_project_bwd(hipStream_t s, const GutParams& P, const uint32_t* tiles_count, const float* density12, const float* sph,
                        const float* rgb, const float* g_rgb, float* g_density12, float* g_sph) {
    hipLaunchKernelGGL(gut_project_bwd_kernel, dim3(div_up(P.N, 128)), dim3(128), 0, s, P, tiles_count,
                       reinterpret_cast<const float4*>(density12), sph, rgb, g_rgb, g_density12, g_sph);
}

}  // namespace grut
