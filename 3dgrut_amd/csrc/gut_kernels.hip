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
// expansion pass on identical stored inputs; kept out-of-line (noinline) so that both kernels run the
// very same instruction sequence and their results agree bit for bit.
__device__ __noinline__ float tile_min_power(float tx, float ty, float4 co, float mx, float my) {
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

// ---------------------------------------------------------------------------------------------
// K1: projection onto tiles — GUTProjector::eval (gutProjector.cuh:217-322)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gut_project_kernel(GutParams P, const float4* __restrict__ density12,
                                                          const float* __restrict__ sph, GutProjected out,
                                                          int32_t* __restrict__ visibility, uint32_t* __restrict__ num_visible) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool has_tiles = false;
    if (i < P.N) {
        const float4 a = density12[3 * (size_t)i + 0];  // pos.xyz, density
        const float4 b = density12[3 * (size_t)i + 1];  // quat wxyz
        const float4 c = density12[3 * (size_t)i + 2];  // scale.xyz, pad
        const f3 pos = mk3(a.x, a.y, a.z);
        const float opacity = a.w;

        uint32_t ntiles = 0;
        int vis = 0;
        float cx = 0.f, cy = 0.f, ex = 0.f, ey = 0.f, depth = 0.f;
        float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
        const float view_z = fmaf(P.poses.view_R[6], pos.x, fmaf(P.poses.view_R[7], pos.y, fmaf(P.poses.view_R[8], pos.z, P.poses.view_t[2])));
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
                            const TileBBox bb = tile_space_bbox(P.gx, P.gy, cx, cy, ex, ey);
                            if (P.tile_culling) {
                                const float pmax2 = logf(co.w / P.min_alpha);  // same expression as the expansion pass
                                for (int y = bb.miny; y < bb.maxy; ++y)
                                    for (int x = bb.minx; x < bb.maxx; ++x)
                                        if (tile_min_power((float)x, (float)y, co, cx, cy) < pmax2) ntiles++;
                            } else {
                                ntiles = (uint32_t)((bb.maxx - bb.minx) * (bb.maxy - bb.miny));
                            }
                        }
                    }
                }
            }
        }
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
            depth = P.global_z ? view_z : dist;
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
    if (r >= P.N) return;
    uint32_t off = r == 0 ? 0u : offsets[r - 1];
    uint32_t max_off = offsets[r];
    if (max_off == off) return;
    if (max_off > capacity) max_off = capacity;
    const uint32_t p = rank_to_particle[r];
    const float2 ext = proj.extent[p];
    if (!(ext.x <= 1e-06f)) {
        const float2 c = proj.proj_pos[p];
        const TileBBox bb = tile_space_bbox(P.gx, P.gy, c.x, c.y, ext.x, ext.y);
        if (P.tile_culling) {
            const float4 co = proj.conic_opacity[p];
            const float pmax = logf(co.w / P.min_alpha);
            for (int y = bb.miny; (y < bb.maxy) && (off < max_off); ++y)
                for (int x = bb.minx; (x < bb.maxx) && (off < max_off); ++x)
                    if (tile_min_power((float)x, (float)y, co, c.x, c.y) < pmax) {
                        tile_keys[off] = (uint32_t)(y * P.gx + x);
                        tile_vals[off] = p;
                        off++;
                    }
        } else {
            for (int y = bb.miny; (y < bb.maxy) && (off < max_off); ++y)
                for (int x = bb.minx; (x < bb.maxx) && (off < max_off); ++x) {
                    tile_keys[off] = (uint32_t)(y * P.gx + x);
                    tile_vals[off] = p;
                    off++;
                }
        }
    }
    for (; off < max_off; ++off) {  // gutProjector.cuh:372-376 padding
        tile_keys[off] = 0xFFFFFFFFu;
        tile_vals[off] = 0xFFFFFFFFu;
    }
}

// ---------------------------------------------------------------------------------------------
// K6: tile ranges — computeSortedTileRangeIndices (gutRenderer.cu:46-76)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gut_tile_ranges_kernel(uint32_t n, uint32_t tile_mask, uint32_t num_tiles,
                                                              const uint32_t* __restrict__ sorted_tile_keys, uint2* __restrict__ ranges) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t t = sorted_tile_keys[k] & tile_mask;
    const bool valid = t < num_tiles;
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
// ---------------------------------------------------------------------------------------------
template <int DEG>
__global__ __launch_bounds__(64) void gut_render_fwd_kernel(GutParams P, const uint2* __restrict__ ranges,
                                                            const uint32_t* __restrict__ sorted_idx,
                                                            const float4* __restrict__ density12, const float* __restrict__ rgb,
                                                            const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                            float4* __restrict__ out_fd, float* __restrict__ out_dist,
                                                            float* __restrict__ out_cnt) {
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

    for (uint32_t b = range.x; b < range.y; b += 64) {
        if (!__any(alive)) break;
        {   // stage 64 entries
            const uint32_t e = b + lane;
            float4 q0, q1, q2, q3, q4;
            q0 = q1 = q2 = make_float4(0.f, 0.f, 0.f, 0.f);
            q3 = make_float4(1.f, 1.f, 1.f, 0.f);
            q4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < range.y) {
                const uint32_t idx = sorted_idx[e];
                if (idx != 0xFFFFFFFFu) {
                    const float4 a = density12[3 * (size_t)idx + 0];
                    const float4 q = density12[3 * (size_t)idx + 1];
                    const float4 s = density12[3 * (size_t)idx + 2];
                    const m3 rt = quat_wxyz_to_rotT(q.x, q.y, q.z, q.w);
                    const float ix = 1.f / s.x, iy = 1.f / s.y, iz = 1.f / s.z;
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
        const int n = (int)min(64u, range.y - b);
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
//   q4 = quat wxyz      | q5 = rgb(clamped), as_float(idx) | q6 = 1/scale.xyz, -
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void quat_bw(f3 p, f3 g, float4 q, float& dr, float& dx, float& dy, float& dz) {
    // matmul_bw_quat (mathUtils.cuh:458-521): accumulates into dr,dx,dy,dz
    const f3 d0 = p * g.x, d1 = p * g.y, d2 = p * g.z;
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    dy += -4.f * y * d0.x; dz += -4.f * z * d0.x;
    dr += 2.f * z * d0.y; dx += 2.f * y * d0.y; dy += 2.f * x * d0.y; dz += 2.f * r * d0.y;
    dr += -2.f * y * d0.z; dx += 2.f * z * d0.z; dy += -2.f * r * d0.z; dz += 2.f * x * d0.z;
    dr += -2.f * z * d1.x; dx += 2.f * y * d1.x; dy += 2.f * x * d1.x; dz += -2.f * r * d1.x;
    dx += -4.f * x * d1.y; dz += -4.f * z * d1.y;
    dr += 2.f * x * d1.z; dx += 2.f * r * d1.z; dy += 2.f * z * d1.z; dz += 2.f * y * d1.z;
    dr += 2.f * y * d2.x; dx += 2.f * z * d2.x; dy += 2.f * r * d2.x; dz += 2.f * x * d2.x;
    dr += -2.f * x * d2.y; dx += -2.f * r * d2.y; dy += 2.f * z * d2.y; dz += 2.f * y * d2.y;
    dx += -4.f * x * d2.z; dy += -4.f * y * d2.z;
}

template <int DEG>
__global__ __launch_bounds__(64) void gut_render_bwd_kernel(GutParams P, const uint2* __restrict__ ranges,
                                                            const uint32_t* __restrict__ sorted_idx,
                                                            const float4* __restrict__ density12, const float* __restrict__ rgb,
                                                            const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                            const float4* __restrict__ fd, const float4* __restrict__ g_fd,
                                                            const float* __restrict__ dist, const float* __restrict__ g_dist,
                                                            float* __restrict__ g_density12, float* __restrict__ g_rgb) {
    __shared__ float4 s_rec[64 * 7];
    __shared__ float4 s_acc[64 * 4];  // per staged entry: 14 reduced gradient terms (+2 pad)
    uint32_t tile, strip;
    if (!strip_mapping(blockIdx.x, P.gx * P.gy, tile, strip)) return;
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
        D_fin = dist[pix]; gD = g_dist[pix];
    }
    const uint2 range = ranges[tile];

    for (uint32_t b = range.x; b < range.y; b += 64) {
        if (!__any(alive)) break;
        {
            const uint32_t e = b + lane;
            float4 q0, q1, q2, q3, q4, q5, q6;
            q0 = q1 = q2 = q4 = make_float4(0.f, 0.f, 0.f, 0.f);
            q3 = make_float4(1.f, 1.f, 1.f, 0.f);
            q5 = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
            q6 = make_float4(1.f, 1.f, 1.f, 0.f);
            if (e < range.y) {
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
                    q4 = q;
                    q5 = make_float4(fmaxf(rgb[3 * (size_t)idx], 0.f), fmaxf(rgb[3 * (size_t)idx + 1], 0.f),
                                     fmaxf(rgb[3 * (size_t)idx + 2], 0.f), __uint_as_float(idx));
                    q6 = make_float4(1.f / s.x, 1.f / s.y, 1.f / s.z, 0.f);
                }
            }
            float4* r = &s_rec[lane * 7];
            r[0] = q0; r[1] = q1; r[2] = q2; r[3] = q3; r[4] = q4; r[5] = q5; r[6] = q6;
        }
        __syncthreads();
        const int n = (int)min(64u, range.y - b);
        unsigned long long hit_entries = 0ull;  // wave-uniform: staged entries with >= 1 hit in this wave
        for (int j = 0; j < n; ++j) {
            if (!__any(alive)) break;
            float g_px = 0.f, g_py = 0.f, g_pz = 0.f, g_dn = 0.f, g_qr = 0.f, g_qx = 0.f, g_qy = 0.f, g_qz = 0.f;
            float g_sx = 0.f, g_sy = 0.f, g_sz = 0.f, g_cr = 0.f, g_cg = 0.f, g_cb = 0.f;
            bool hit = false;
            if (alive) {
                const float4* rec = &s_rec[j * 7];
                const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q6 = rec[6];
                const m3 rotT = {mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q2.x, q2.y, q2.z)};
                const f3 gscl = mk3(q3.x, q3.y, q3.z), giscl = mk3(q6.x, q6.y, q6.z);
                const float dens = q3.w;
                const f3 gposc = ray.o - mk3(q0.w, q1.w, q2.w);
                const f3 gposcr = mul_rows(rotT, gposc);
                const f3 gro = giscl * gposcr;
                const f3 rdr = mul_rows(rotT, ray.d);
                const f3 grdu = giscl * rdr;
                const float l2 = dot(grdu, grdu);
                const float il = l2 > 0.f ? __builtin_amdgcn_rsqf(l2) : 1.f;  // safe_normalize
                const f3 grd = grdu * il;
                const f3 gcrod = cross(grd, gro);
                const float gray = dot(gcrod, gcrod);
                const float gres = particle_response<DEG>(gray);
                const float galpha = fminf(P.max_alpha, gres * dens);
                if ((gres > P.min_response) && (galpha > P.min_alpha)) {
                    hit = true;
                    const float4 q4 = rec[4], q5 = rec[5];
                    const f3 feat = mk3(q5.x, q5.y, q5.z);
                    const float pdot = -dot(grd, gro);
                    const f3 grdd = grd * pdot;
                    const f3 grds = gscl * grdd;
                    const float gsq = dot(grds, grds);
                    const float gdist = __builtin_amdgcn_sqrtf(gsq);
                    const float weight = galpha * T;
                    const float nextT = (1.f - galpha) * T;
                    const float inextT = nextT <= P.min_transmittance ? 0.f : __builtin_amdgcn_rcpf(nextT);

                    D = fmaf(weight, gdist, D);
                    const float resHitT = fmaxf((D_fin - D) * inextT, 0.f);
                    const float galphaRayHitGrd = (gdist - resHitT) * T * gD;
                    const f3 grdsRayHitGrd = gsq > 0.f ? grds * (weight * __builtin_amdgcn_rcpf(gdist) * gD) : mk3(0.f, 0.f, 0.f);
                    const f3 gsclRayHitGrd = grdd * grdsRayHitGrd;
                    const float grdScaledDot = dot(grdsRayHitGrd * gscl, grd);
                    const f3 grdRayHitGrd = (gscl * grdsRayHitGrd) * pdot - gro * grdScaledDot;
                    const f3 groRayHitGrd = grd * (-grdScaledDot);

                    const float resTrm = galpha < 0.999999f ? T_fin * __builtin_amdgcn_rcpf(1.f - galpha) : T;
                    const float galphaRayDnsGrd = resTrm * -gT;

                    g_cr = gC.x * weight; g_cg = gC.y * weight; g_cb = gC.z * weight;
                    Cr = fmaf(feat.x, weight, Cr); Cg = fmaf(feat.y, weight, Cg); Cb = fmaf(feat.z, weight, Cb);
                    const f3 resRad = mk3(fmaxf((C_fin.x - Cr) * inextT, 0.f), fmaxf((C_fin.y - Cg) * inextT, 0.f),
                                          fmaxf((C_fin.z - Cb) * inextT, 0.f));
                    const float common = galphaRayHitGrd + galphaRayDnsGrd +
                                         T * ((feat.x - resRad.x) * gC.x + (feat.y - resRad.y) * gC.y + (feat.z - resRad.z) * gC.z);
                    g_dn = gres * common;
                    const float gresGrd = dens * common;
                    const float grayGrd = particle_response_grd<DEG>(gray, gres, gresGrd);

                    const f3 gcrodGrd = gcrod * (2.f * grayGrd);
                    const f3 grdGrd = mk3(gcrodGrd.z * gro.y - gcrodGrd.y * gro.z, gcrodGrd.x * gro.z - gcrodGrd.z * gro.x,
                                          gcrodGrd.y * gro.x - gcrodGrd.x * gro.y);
                    const f3 groGrd = mk3(gcrodGrd.y * grd.z - gcrodGrd.z * grd.y, gcrodGrd.z * grd.x - gcrodGrd.x * grd.z,
                                          gcrodGrd.x * grd.y - gcrodGrd.y * grd.x);
                    const f3 groTot = groGrd + groRayHitGrd;
                    // d gro / d scale = -gposcr / scale^2 = -gro / scale
                    const f3 gsclGrdGro = mk3(-gro.x * giscl.x, -gro.y * giscl.y, -gro.z * giscl.z) * groTot;
                    const f3 gposcrGrd = giscl * groTot;
                    const f3 gposcGrd = mul_cols(rotT, gposcrGrd);
                    g_px = -gposcGrd.x; g_py = -gposcGrd.y; g_pz = -gposcGrd.z;
                    quat_bw(gposc, gposcrGrd, q4, g_qr, g_qx, g_qy, g_qz);

                    // safe_normalize_bw(grdu, grdGrd + grdRayHitGrd)
                    const f3 gsum = grdGrd + grdRayHitGrd;
                    f3 grduGrd = mk3(0.f, 0.f, 0.f);
                    if (l2 > 0.f) {
                        const float il3 = il * il * il;
                        const float sdot = gsum.x * grdu.x + gsum.y * grdu.y + gsum.z * grdu.z;
                        grduGrd = gsum * il - grdu * (il3 * sdot);
                    }
                    const f3 sclFromDir = mk3(-grdu.x * giscl.x, -grdu.y * giscl.y, -grdu.z * giscl.z) * grduGrd;
                    g_sx = gsclRayHitGrd.x + gsclGrdGro.x + sclFromDir.x;
                    g_sy = gsclRayHitGrd.y + gsclGrdGro.y + sclFromDir.y;
                    g_sz = gsclRayHitGrd.z + gsclGrdGro.z + sclFromDir.z;
                    const f3 rdrGrd = giscl * grduGrd;
                    quat_bw(ray.d, rdrGrd, q4, g_qr, g_qx, g_qy, g_qz);

                    T = nextT;
                    if (T < P.min_transmittance) alive = false;
                }
            }
            if (__any(hit)) {
                hit_entries |= (1ull << j);
                g_px = wave_sum_to_lane63(g_px); g_py = wave_sum_to_lane63(g_py); g_pz = wave_sum_to_lane63(g_pz);
                g_dn = wave_sum_to_lane63(g_dn);
                g_qr = wave_sum_to_lane63(g_qr); g_qx = wave_sum_to_lane63(g_qx); g_qy = wave_sum_to_lane63(g_qy); g_qz = wave_sum_to_lane63(g_qz);
                g_sx = wave_sum_to_lane63(g_sx); g_sy = wave_sum_to_lane63(g_sy); g_sz = wave_sum_to_lane63(g_sz);
                g_cr = wave_sum_to_lane63(g_cr); g_cg = wave_sum_to_lane63(g_cg); g_cb = wave_sum_to_lane63(g_cb);
                if (lane == 63) {
                    s_acc[j * 4 + 0] = make_float4(g_px, g_py, g_pz, g_dn);
                    s_acc[j * 4 + 1] = make_float4(g_qr, g_qx, g_qy, g_qz);
                    s_acc[j * 4 + 2] = make_float4(g_sx, g_sy, g_sz, 0.f);
                    s_acc[j * 4 + 3] = make_float4(g_cr, g_cg, g_cb, 0.f);
                }
            }
        }
        __syncthreads();
        // flush: lane j owns staged entry j; one atomic set per (strip, particle with a hit)
        if ((hit_entries >> lane) & 1ull) {
            const uint32_t idx = __float_as_uint(s_rec[lane * 7 + 5].w);
            const float4 a0 = s_acc[lane * 4 + 0], a1 = s_acc[lane * 4 + 1], a2 = s_acc[lane * 4 + 2], a3 = s_acc[lane * 4 + 3];
            float* gd = g_density12 + 12 * (size_t)idx;
            atomicAdd(gd + 0, a0.x); atomicAdd(gd + 1, a0.y); atomicAdd(gd + 2, a0.z); atomicAdd(gd + 3, a0.w);
            atomicAdd(gd + 4, a1.x); atomicAdd(gd + 5, a1.y); atomicAdd(gd + 6, a1.z); atomicAdd(gd + 7, a1.w);
            atomicAdd(gd + 8, a2.x); atomicAdd(gd + 9, a2.y); atomicAdd(gd + 10, a2.z);
            float* gr = g_rgb + 3 * (size_t)idx;
            atomicAdd(gr + 0, a3.x); atomicAdd(gr + 1, a3.y); atomicAdd(gr + 2, a3.z);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// K9: projection backward — GUTProjector::evalBackward (gutProjector.cuh:390-430): per visible particle
// dRGB -> dSH (clamp-masked) and d direction -> d position; every SH gradient row is written exactly once
// (zeros for particles without tiles), so the caller does not need to zero-fill grad_sph.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gut_project_bwd_kernel(GutParams P, const uint32_t* __restrict__ tiles_count,
                                                              const float4* __restrict__ density12, const float* __restrict__ sph,
                                                              const float* __restrict__ rgb, const float* __restrict__ g_rgb,
                                                              float* __restrict__ g_density12, float* __restrict__ g_sph) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.N) return;
    float* gs = g_sph + (size_t)i * 3 * P.ncoef;
    const int nact = min((P.n_active + 1) * (P.n_active + 1), P.ncoef);
    if (tiles_count[i] == 0) {
        for (int k = 0; k < 3 * P.ncoef; ++k) gs[k] = 0.f;
        return;
    }
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
    const float* coef = sph + (size_t)i * 3 * P.ncoef;
    f3 gdir = mk3(0.f, 0.f, 0.f);
    for (int k = 0; k < P.ncoef; ++k) {
        if (k < nact) {
            gs[3 * k] = basis[k] * g.x; gs[3 * k + 1] = basis[k] * g.y; gs[3 * k + 2] = basis[k] * g.z;
            const float s = g.x * coef[3 * k] + g.y * coef[3 * k + 1] + g.z * coef[3 * k + 2];
            gdir = gdir + dbasis[k] * s;
        } else {
            gs[3 * k] = 0.f; gs[3 * k + 1] = 0.f; gs[3 * k + 2] = 0.f;
        }
    }
    const float ng = dot(dir, gdir);
    const f3 gpos = (gdir - dir * ng) * ilen;
    g_density12[12 * (size_t)i + 0] += gpos.x;
    g_density12[12 * (size_t)i + 1] += gpos.y;
    g_density12[12 * (size_t)i + 2] += gpos.z;
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
                        uint32_t* ranges) {
    hipLaunchKernelGGL(gut_tile_ranges_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, tile_mask, num_tiles, sorted_tile_keys,
                       reinterpret_cast<uint2*>(ranges));
}

static uint32_t strip_grid(const GutParams& P) {
    const uint32_t tiles = (uint32_t)(P.gx * P.gy);
    return ((tiles + 7u) & ~7u) * 4u;
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
                       const float* rgb, const float* ray_o, const float* ray_d, float* out_fd, float* out_dist, float* out_cnt) {
    GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL(gut_render_fwd_kernel<D_>, dim3(strip_grid(P)), dim3(64), 0, s, P,
                                                      reinterpret_cast<const uint2*>(ranges), sorted_idx,
                                                      reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d,
                                                      reinterpret_cast<float4*>(out_fd), out_dist, out_cnt));
}
void launch_render_bwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_idx, const float* density12,
                       const float* rgb, const float* ray_o, const float* ray_d, const float* fd, const float* g_fd, const float* dist,
                       const float* g_dist, float* g_density12, float* g_rgb) {
    GRUT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL(gut_render_bwd_kernel<D_>, dim3(strip_grid(P)), dim3(64), 0, s, P,
                                                      reinterpret_cast<const uint2*>(ranges), sorted_idx,
                                                      reinterpret_cast<const float4*>(density12), rgb, ray_o, ray_d,
                                                      reinterpret_cast<const float4*>(fd), reinterpret_cast<const float4*>(g_fd), dist,
                                                      g_dist, g_density12, g_rgb));
}
void launch_project_bwd(hipStream_t s, const GutParams& P, const uint32_t* tiles_count, const float* density12, const float* sph,
                        const float* rgb, const float* g_rgb, float* g_density12, float* g_sph) {
    hipLaunchKernelGGL(gut_project_bwd_kernel, dim3(div_up(P.N, 256)), dim3(256), 0, s, P, tiles_count,
                       reinterpret_cast<const float4*>(density12), sph, rgb, g_rgb, g_density12, g_sph);
}

}  // namespace grut
