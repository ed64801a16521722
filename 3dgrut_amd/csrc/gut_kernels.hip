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
    const FramePoses& FP = frame_poses(P);
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

        const float view_z = fmaf(FP.view_R[6], pos.x, fmaf(FP.view_R[7], pos.y, fmaf(FP.view_R[8], pos.z, FP.view_t[2])));
        view_z_keep = view_z;
        bool ok = (opacity >= P.min_alpha) && (view_z >= 0.2f);
        if (ok) {
            const m3 rotT = quat_wxyz_to_rotT(b.x, b.y, b.z, b.w);
            float spx[7], spy[7];
            int nvalid = 0;
            nvalid += project_point_with_shutter(P.cam, FP, P.n_rs_iter, pos, P.ut_margin, spx[0], spy[0]) ? 1 : 0;
            cx = spx[0] * P.ut_w0m; cy = spy[0] * P.ut_w0m;
            const f3 axes[3] = {rotT.r0 * (P.ut_delta * c.x), rotT.r1 * (P.ut_delta * c.y), rotT.r2 * (P.ut_delta * c.z)};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                nvalid += project_point_with_shutter(P.cam, FP, P.n_rs_iter, pos + axes[k], P.ut_margin, spx[k + 1], spy[k + 1]) ? 1 : 0;
                cx += P.ut_wi * spx[k + 1]; cy += P.ut_wi * spy[k + 1];
                nvalid += project_point_with_shutter(P.cam, FP, P.n_rs_iter, pos - axes[k], P.ut_margin, spx[k + 4], spy[k + 4]) ? 1 : 0;
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
            const f3 ray = pos - mk3(FP.s2w_t[0], FP.s2w_t[1], FP.s2w_t[2]);
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
                                                         uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ tile_vals,
                                                         uint32_t* __restrict__ pos_particle) {
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
            proj.part_offset[p] = off;
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
                           if (keep && slot < send) {  // the sort carries the expansion position, not the particle
                               tile_keys[slot] = tile;
                               tile_vals[slot] = slot;
                               pos_particle[slot] = sp;
                           }
                           o += (uint32_t)__popcll(m);
                       });
        for (uint32_t k = o + lane; k < send; k += 64) {  // gutProjector.cuh:372-376 padding
            tile_keys[k] = 0xFFFFFFFFu;
            tile_vals[k] = k;
            pos_particle[k] = 0xFFFFFFFFu;
        }
    }
}

__global__ __launch_bounds__(256) void gut_gather_particle_idx_kernel(uint32_t n, const uint32_t* __restrict__ sorted_pos,
                                                                      const uint32_t* __restrict__ pos_particle, uint32_t* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = pos_particle[sorted_pos[k]];
}

// ---------------------------------------------------------------------------------------------
// K6: tile ranges — computeSortedTileRangeIndices (gutRenderer.cu:46-76)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gut_tile_ranges_kernel(uint32_t n_cap, const uint32_t* __restrict__ n_dev, uint32_t tile_mask,
                                                              uint32_t num_tiles, const uint32_t* __restrict__ sorted_tile_keys,
                                                              uint2* __restrict__ ranges, uint32_t* __restrict__ boundary_tile) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = n_dev ? min(n_cap, *n_dev) : n_cap;  // n_cap may be a capacity bound (speculative launch)
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
constexpr uint32_t kGatherSmall = 128;  // tile counts up to this are gathered by the owning lane
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

template <int STRIDE>
__device__ __forceinline__ void add_slot(const float* __restrict__ partial, size_t slot, float (&acc)[STRIDE]) {
    const float4* pp = reinterpret_cast<const float4*>(partial + slot * STRIDE);
#pragma unroll
    for (int k = 0; k < STRIDE / 4; ++k) {
        const float4 v = pp[k];
        acc[4 * k] += v.x; acc[4 * k + 1] += v.y; acc[4 * k + 2] += v.z; acc[4 * k + 3] += v.w;
    }
}

template <int STRIDE>
__global__ __launch_bounds__(256) void gut_grad_gather_kernel(GutParams P, GutProjected proj, const float4* __restrict__ density12,
                                                              GutGradSlots slots, int have_partials, float* __restrict__ g_density12,
                                                              float* __restrict__ g_rgb) {
    const int lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t count = (i < P.N) ? proj.tiles_count[i] : 0u;
    const bool has = count != 0;
    float acc[STRIDE];
#pragma unroll
    for (int k = 0; k < STRIDE; ++k) acc[k] = 0.f;
    const uint32_t off = has ? proj.part_offset[i] : 0u;
    if (have_partials) {
        if (has && count <= kGatherSmall) {
            // 16 flags per (unaligned) load; set flags are rare (most tile entries lie behind the rays' termination)
            const size_t s0 = 2 * (size_t)off;
            const uint32_t nb = 2 * count;
            for (uint32_t c = 0; c < nb; c += 16) {
                const FlagChunk w = *reinterpret_cast<const FlagChunk*>(slots.flag + s0 + c);  // buffer has 32 B of slack
                const uint32_t rem = nb - c;
                unsigned long long lo = w.lo, hi = w.hi;
                if (rem < 8) { lo &= (1ull << (8 * rem)) - 1ull; hi = 0ull; }
                else if (rem < 16) hi &= (rem == 8) ? 0ull : ((1ull << (8 * (rem - 8))) - 1ull);
                while (lo) {
                    const int b = (__ffsll((long long)lo) - 1) >> 3;
                    lo &= ~(0xFFull << (8 * b));
                    add_slot<STRIDE>(slots.partial, s0 + c + b, acc);
                }
                while (hi) {
                    const int b = (__ffsll((long long)hi) - 1) >> 3;
                    hi &= ~(0xFFull << (8 * b));
                    add_slot<STRIDE>(slots.partial, s0 + c + 8 + b, acc);
                }
            }
        }
        unsigned long long big = __ballot(has && count > kGatherSmall);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const size_t s0 = 2 * (size_t)(uint32_t)__builtin_amdgcn_readlane((int)off, src);
            const uint32_t n2 = 2u * (uint32_t)__builtin_amdgcn_readlane((int)count, src);
            float part[STRIDE];
#pragma unroll
            for (int k = 0; k < STRIDE; ++k) part[k] = 0.f;
            for (uint32_t k = lane; k < n2; k += 64)
                if (slots.flag[s0 + k]) add_slot<STRIDE>(slots.partial, s0 + k, part);
#pragma unroll
            for (int k = 0; k < STRIDE; ++k) {
                const float tot = wave_sum(part[k]);
                if (lane == src) acc[k] = tot;
            }
        }
    }
    if (i >= P.N) return;
    float4* gd = reinterpret_cast<float4*>(g_density12 + 12 * (size_t)i);
    if (!has) {
        gd[0] = gd[1] = gd[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4 q = density12[3 * (size_t)i + 1], sc = density12[3 * (size_t)i + 2];
    const m3 rt = quat_wxyz_to_rotT(q.x, q.y, q.z, q.w);
    const f3 B = mk3(acc[0], acc[1], acc[2]);
    const float* m = &acc[4];
    // position = -R B  (matmul_bw_vec with rows of R^T); the SH direction term is added by the projection backward
    const f3 gpos = mk3(-(B.x * rt.r0.x + B.y * rt.r1.x + B.z * rt.r2.x), -(B.x * rt.r0.y + B.y * rt.r1.y + B.z * rt.r2.y),
                        -(B.x * rt.r0.z + B.y * rt.r1.z + B.z * rt.r2.z));
    // scale_i = -(1/s_i) sum_j rotT_ij m_ij (+ direct hit-distance term)
    float gsx = -(rt.r0.x * m[0] + rt.r0.y * m[1] + rt.r0.z * m[2]) / sc.x;
    float gsy = -(rt.r1.x * m[3] + rt.r1.y * m[4] + rt.r1.z * m[5]) / sc.y;
    float gsz = -(rt.r2.x * m[6] + rt.r2.y * m[7] + rt.r2.z * m[8]) / sc.z;
    if (STRIDE > 16) { gsx += acc[16]; gsy += acc[17]; gsz += acc[18]; }
    const float4 dq = quat_contract(m, make_float4(2.f * q.x, 2.f * q.y, 2.f * q.z, 2.f * q.w));
    gd[0] = make_float4(gpos.x, gpos.y, gpos.z, acc[3]);
    gd[1] = dq;
    gd[2] = make_float4(gsx, gsy, gsz, 0.f);
    g_rgb[3 * (size_t)i] = acc[13];
    g_rgb[3 * (size_t)i + 1] = acc[14];
    g_rgb[3 * (size_t)i + 2] = acc[15];
}

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
    const FramePoses& FP = frame_poses(P);
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
        const f3 v = mk3(a.x, a.y, a.z) - mk3(FP.s2w_t[0], FP.s2w_t[1], FP.s2w_t[2]);
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

// frame poses from camera-to-world matrices in device memory (no host round trip, no stream sync)
__global__ void gut_frame_poses_kernel(const float* __restrict__ T_start, const float* __restrict__ T_end, FramePoses* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float ps[7], pe[7];
    c2w_to_world_to_sensor(T_start, ps);
    if (T_end) c2w_to_world_to_sensor(T_end, pe);
    else
        for (int k = 0; k < 7; ++k) pe[k] = ps[k];
    *out = make_frame_poses(ps, pe);
}

}  // namespace

void launch_frame_poses(hipStream_t s, const float* T_start, const float* T_end, FramePoses* out) {
    hipLaunchKernelGGL(gut_frame_poses_kernel, dim3(1), dim3(64), 0, s, T_start, T_end, out);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void launch_project(hipStream_t s, const GutParams& P, const float* density12, const float* sph, const GutProjected& out,
                    int32_t* visibility, uint32_t* num_visible) {
    hipLaunchKernelGGL(gut_project_kernel, dim3(div_up(P.N, 256)), dim3(256), 0, s, P, reinterpret_cast<const float4*>(density12), sph,
                       out, visibility, num_visible);
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
    hipLaunchKernelGGL(gut_tile_ranges_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, n_dev, tile_mask, num_tiles, sorted_tile_keys,
                       reinterpret_cast<uint2*>(ranges), boundary_tile);
}

void launch_project_bwd(hipStream_t s, const GutParams& P, const GutProjected& proj, const float* density12, const float* sph,
                        const float* g_rgb, float* g_density12, float* g_sph) {
    hipLaunchKernelGGL(gut_project_bwd_kernel, dim3(div_up(P.N, 128)), dim3(128), 0, s, P, proj.tiles_count,
                       reinterpret_cast<const float4*>(density12), sph, proj.rgb, g_rgb, g_density12, g_sph);
}
void launch_grad_finalize(hipStream_t s, const GutParams& P, const GutProjected& proj, const float* density12, const float* sph,
                          const GutGradSlots& slots, bool has_gdist, bool have_partials, float* g_rgb, float* g_density12, float* g_sph) {
    const dim3 grid(div_up(P.N, 256)), block(256);
    if (has_gdist)
        hipLaunchKernelGGL(gut_grad_gather_kernel<20>, grid, block, 0, s, P, proj, reinterpret_cast<const float4*>(density12), slots,
                           have_partials ? 1 : 0, g_density12, g_rgb);
    else
        hipLaunchKernelGGL(gut_grad_gather_kernel<16>, grid, block, 0, s, P, proj, reinterpret_cast<const float4*>(density12), slots,
                           have_partials ? 1 : 0, g_density12, g_rgb);
    hipLaunchKernelGGL(gut_project_bwd_kernel, dim3(div_up(P.N, 128)), dim3(128), 0, s, P, proj.tiles_count,
                       reinterpret_cast<const float4*>(density12), sph, proj.rgb, g_rgb, g_density12, g_sph);
}

}  // namespace grut
