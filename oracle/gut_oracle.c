/*
 * gut_oracle.c — CPU restatement of the reference 3DGUT path (threedgut_tracer).
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker; never imported by the product package.
 *
 * Parity pinning: checked against the reference's own sources compiled on
 * the host (oracle/_ref, recipes in oracle/ref/) and against the tests/golden/ vectors they
 * produced — the per-hit math, the camera models, the projection + expansion stage, and the
 * whole frame: the reference's projectOnTiles / render / renderBackward kernels run block by
 * block on a fiber emulation of a CUDA thread block (tests/golden/gut_render.npz: sorted
 * lists bit-identical, images within 2e-6, gradients within 2e-5).  Not pinned: what exists
 * in the reference only as Slang source (the projection backward, the K > 0 per-hit
 * backward).  The reference ships no tests or golden vectors of its own for this path
 * (SURVEY.md §4, §8c).
 *
 * Sequence restated: gutRenderer.cu:241-520 (GUTRenderer::renderForward/Backward).
 */
#include "orc_math.h"
#include "../include/grut_amd.h"

#include <stdio.h>
#include <stdlib.h>

#define ORC_TILE 16 /* gutRendererParameters.h:23-24 BlockX/BlockY */
#define ORC_INVALID_IDX 0xFFFFFFFFu

/* --------------------------------------------------------------------------------------
 * camera projections — cameraProjections.cuh:24-257
 * ------------------------------------------------------------------------------------ */
static real stable_norm2(real x, real y) { /* :25-35 */
    const real ax = r_fabs(x), ay = r_fabs(y);
    const real mn = r_min(ax, ay), mx = r_max(ax, ay);
    if (mx <= 0) return 0;
    const real r = mn / mx;
    return mx * r_sqrt(1 + r * r);
}
static real poly_horner(const float* c, int n, real x) { /* :37-48 */
    real y = (real)c[n - 1];
    for (int i = n - 2; i >= 0; --i) y = x * y + (real)c[i];
    return y;
}
static int within_resolution(real w, real h, real tol, real px, real py) { /* :67-70 */
    const real mx = w * tol, my = h * tol;
    return (px > -mx) && (py > -my) && (px < w + mx) && (py < h + my);
}
static real relative_shutter_time(const GrutCamera* cam, real px, real py) { /* :50-65 */
    switch (cam->shutter) {
    case GRUT_SHUTTER_ROLLING_TOP_TO_BOTTOM: return r_floor(py) / ((real)cam->height - 1);
    case GRUT_SHUTTER_ROLLING_LEFT_TO_RIGHT: return r_floor(px) / ((real)cam->width - 1);
    case GRUT_SHUTTER_ROLLING_BOTTOM_TO_TOP: return ((real)cam->height - r_ceil(py)) / ((real)cam->height - 1);
    case GRUT_SHUTTER_ROLLING_RIGHT_TO_LEFT: return ((real)cam->width - r_ceil(px)) / ((real)cam->width - 1);
    default: return R_(0.5);
    }
}
static int project_pinhole(const GrutCamera* cam, v3 p, real tol, real* ox, real* oy) { /* :72-118 */
    if (p.z <= 0) { *ox = 0; *oy = 0; return 0; }
    const real u = p.x / p.z, v = p.y / p.z;
    const real u2 = u * u, v2 = v * v, r2 = u2 + v2;
    const real a1 = 2 * u * v, a2 = r2 + 2 * u2, a3 = r2 + 2 * v2;
    const float* k = cam->radial;
    const real num = 1 + r2 * ((real)k[0] + r2 * ((real)k[1] + r2 * (real)k[2]));
    const real den = 1 + r2 * ((real)k[3] + r2 * ((real)k[4] + r2 * (real)k[5]));
    const real icD = num / den;
    const real dx = (real)cam->tangential[0] * a1 + (real)cam->tangential[1] * a2 + r2 * ((real)cam->thin_prism[0] + r2 * (real)cam->thin_prism[1]);
    const real dy = (real)cam->tangential[0] * a3 + (real)cam->tangential[1] * a1 + r2 * ((real)cam->thin_prism[2] + r2 * (real)cam->thin_prism[3]);
    const real und_x = icD * u + dx, und_y = icD * v + dy;
    const int valid_radial = (icD > R_(0.8)) && (icD < R_(1.2));
    if (valid_radial) {
        *ox = und_x * (real)cam->focal_length[0] + (real)cam->principal_point[0];
        *oy = und_y * (real)cam->focal_length[1] + (real)cam->principal_point[1];
    } else {
        const real clip = r_hypot((real)cam->width, (real)cam->height);
        *ox = (clip / r_sqrt(r2)) * u + (real)cam->principal_point[0];
        *oy = (clip / r_sqrt(r2)) * v + (real)cam->principal_point[1];
    }
    return valid_radial && within_resolution((real)cam->width, (real)cam->height, tol, *ox, *oy);
}
static int project_fisheye(const GrutCamera* cam, v3 p, real tol, real* ox, real* oy) { /* :120-146 */
    real rho = stable_norm2(p.x, p.y);
    if (rho <= 0) rho = R_(1.1920929e-07);
    const real theta_full = r_atan2(rho, p.z);
    const real theta = r_min(theta_full, (real)cam->max_angle);
    const real t2 = theta * theta;
    const real delta = (theta * (poly_horner(cam->radial, 4, t2) * t2 + 1)) / rho;
    *ox = (real)cam->focal_length[0] * p.x * delta + (real)cam->principal_point[0];
    *oy = (real)cam->focal_length[1] * p.y * delta + (real)cam->principal_point[1];
    return (theta < (real)cam->max_angle) && within_resolution((real)cam->width, (real)cam->height, tol, *ox, *oy);
}
static int project_ftheta(const GrutCamera* cam, v3 p, real tol, real* ox, real* oy) { /* :148-198 */
    real rho = stable_norm2(p.x, p.y);
    if (rho <= 0) rho = R_(1.1920929e-07);
    const real theta_full = r_atan2(rho, p.z);
    const real theta = r_min(theta_full, (real)cam->max_angle);
    real delta;
    if (cam->ftheta_reference_poly == GRUT_FTHETA_PIXELDIST_TO_ANGLE) {
        delta = poly_horner(cam->ftheta_angle_to_pixeldist, 6, theta);
        float dpoly[5];
        for (int i = 1; i < 6; ++i) dpoly[i - 1] = (float)i * cam->ftheta_pixeldist_to_angle[i];
        for (int it = 0; it < 3; ++it) {
            const real dfdx = poly_horner(dpoly, 5, delta);
            const real res  = poly_horner(cam->ftheta_pixeldist_to_angle, 6, delta) - theta;
            delta -= res / dfdx;
        }
    } else {
        delta = poly_horner(cam->ftheta_angle_to_pixeldist, 6, theta);
    }
    const real s = delta / rho;
    *ox = s * ((real)cam->ftheta_linear_cde[0] * p.x + (real)cam->ftheta_linear_cde[1] * p.y);
    *oy = s * ((real)cam->ftheta_linear_cde[2] * p.x + p.y);
    *ox += (real)cam->principal_point[0] + R_(0.5);
    *oy += (real)cam->principal_point[1] + R_(0.5);
    return (theta < (real)cam->max_angle) && within_resolution((real)cam->width, (real)cam->height, tol, *ox, *oy);
}
static int project_point(const GrutCamera* cam, v3 p, real tol, real* ox, real* oy) { /* :200-216 */
    switch (cam->model) {
    case GRUT_CAMERA_OPENCV_PINHOLE: return project_pinhole(cam, p, tol, ox, oy);
    case GRUT_CAMERA_OPENCV_FISHEYE: return project_fisheye(cam, p, tol, ox, oy);
    case GRUT_CAMERA_FTHETA: return project_ftheta(cam, p, tol, ox, oy);
    default: *ox = 0; *oy = 0; return 0;
    }
}
static v3 pose_apply(orc_pose p, v3 x) { const m33 R = quat_xyzw_to_R(p.q); return v3_add(m33_apply(&R, x), p.t); }

/* :218-257 projectPointWithShutter */
static int project_point_with_shutter(const GrutCamera* cam, orc_pose ps, orc_pose pe, int n_iter, v3 x, real tol, real* ox, real* oy) {
    int valid = project_point(cam, pose_apply(ps, x), tol, ox, oy);
    if (cam->shutter == GRUT_SHUTTER_GLOBAL) return valid;
    if (!valid) {
        valid = project_point(cam, pose_apply(pe, x), tol, ox, oy);
        if (!valid) return 0;
    }
    for (int i = 0; i < n_iter; ++i) {
        const real a = relative_shutter_time(cam, *ox, *oy);
        valid = project_point(cam, pose_apply(pose_interpolate(ps, pe, a), x), tol, ox, oy);
    }
    return valid;
}

/* --------------------------------------------------------------------------------------
 * projection onto tiles — gutProjector.cuh:32-322
 * ------------------------------------------------------------------------------------ */
typedef struct { int minx, miny, maxx, maxy; } tile_bbox;

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* :32-43 computeTileSpaceBBox */
static tile_bbox tile_space_bbox(int gx, int gy, real px, real py, real ex, real ey) {
    tile_bbox b;
    b.minx = clampi((int)r_floor((px - R_(0.5) - ex) / ORC_TILE), 0, gx);
    b.miny = clampi((int)r_floor((py - R_(0.5) - ey) / ORC_TILE), 0, gy);
    b.maxx = clampi((int)r_ceil((px - R_(0.5) + ex) / ORC_TILE), 0, gx);
    b.maxy = clampi((int)r_ceil((py - R_(0.5) + ey) / ORC_TILE), 0, gy);
    return b;
}
static real saturate(real x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }
static real copysign_r(real mag, real sgn) { return (real)copysign((double)mag, (double)sgn); }

/* :49-78 tileMinParticlePowerResponse */
static real tile_min_power(real tx, real ty, const real co[4], real mx, real my) {
    const real ts = ORC_TILE;
    const real tminx = ts * tx, tminy = ts * ty, tmaxx = ts + tminx, tmaxy = ts + tminy;
    const real offx = tminx - mx, offy = tminy - my;
    const real lax = offx > 0 ? 1 : 0, lay = offy > 0 ? 1 : 0;
    const real nrx = lax + (mx > tmaxx ? 1 : 0), nry = lay + (my > tmaxy ? 1 : 0);
    if ((nrx + nry) > 0) {
        /* mix(tileMax, tileMin, leftAbove) = tileMax*(1-a) + tileMin*a */
        const real px = tmaxx * (1 - lax) + tminx * lax, py = tmaxy * (1 - lay) + tminy * lay;
        const real dx = copysign_r(ts, offx), dy = copysign_r(ts, offy);
        const real diffx = mx - px, diffy = my - py;
        const real rcpx = 1 / (ts * ts * co[0]), rcpy = 1 / (ts * ts * co[2]);
        const real tx_ = nry * saturate((dx * co[0] * diffx + dx * co[1] * diffy) * rcpx);
        const real ty_ = nrx * saturate((dy * co[1] * diffx + dy * co[2] * diffy) * rcpy);
        const real mdx = mx - (px + tx_ * dx), mdy = my - (py + ty_ * dy);
        return R_(0.5) * (co[0] * mdx * mdx + co[2] * mdy * mdy) + co[1] * mdx * mdy;
    }
    return 0;
}

/* :81-116 computeProjectedExtentConicOpacity */
static int extent_conic_opacity(const GutConfig* cfg, const real cov[3], real opacity, real ext[2], real co[4], real* power_max) {
    const real dil = R_(0.3); /* threedgut.cuh:58 CovarianceDilation */
    const real a = cov[0] + dil, b = cov[1], c = cov[2] + dil;
    const real det = a * c - b * b;
    if (det == 0) return 0;
    co[0] = c / det; co[1] = -b / det; co[2] = a / det;
    const real cov_det = cov[0] * cov[2] - cov[1] * cov[1];
    co[3] = opacity * r_sqrt(r_max(R_(0.000025), cov_det / det)); /* MipSplattingScaling=true (threedgut.cuh:67) */
    const real thr = (real)cfg->particle_kernel_min_alpha;
    if (co[3] < thr) return 0;
    *power_max = r_log(co[3] / thr);
    const real ef = cfg->tight_opacity_bounding ? r_min(R_(3.33), r_sqrt(2 * (*power_max))) : R_(3.33);
    const real mid = R_(0.5) * (a + c);
    const real lambda = mid + r_sqrt(r_max(R_(0.01), mid * mid - det));
    const real radius = ef * r_sqrt(lambda);
    if (cfg->rect_bounding) {
        ext[0] = r_min(ef * r_sqrt(a), radius);
        ext[1] = r_min(ef * r_sqrt(c), radius);
    } else {
        ext[0] = radius; ext[1] = radius;
    }
    return radius > 0;
}

typedef struct {
    orc_pose start, end, mid, mid_inv;
    v3 sensor_world_pos;
    m33 view_R;   /* world->sensor rotation (rows) of the mid pose */
    v3 view_t;
    m33 s2w_R;    /* sensor->world rotation from the quaternion of the inverted pose */
    v3 s2w_t;
} orc_frame_poses;

/* gutRenderer.cu:266-267, 282-284, 407 */
static orc_frame_poses frame_poses(const real* pose_start7, const real* pose_end7) {
    orc_frame_poses f;
    f.start = pose_from7(pose_start7);
    f.end = pose_from7(pose_end7);
    f.mid = pose_interpolate(f.start, f.end, R_(0.5));
    f.mid_inv = pose_inverse(f.mid);
    f.sensor_world_pos = f.mid_inv.t;
    f.view_R = quat_xyzw_to_R(f.mid.q);
    f.view_t = f.mid.t;
    f.s2w_R = quat_xyzw_to_R(f.mid_inv.q);
    f.s2w_t = f.mid_inv.t;
    return f;
}

static int tile_grid_dim(int n) { return (n + ORC_TILE - 1) / ORC_TILE; }

/* GUTProjector::eval, gutProjector.cuh:217-322 (+ unscentedParticleProjection :118-215) for all particles.
 * Quirk: the reference writes visibility = validConicEstimation computed from an UNINITIALISED covariance when the
 * projection itself was rejected early (:131-140 return before :199); here visibility = validProjection && validConic. */
int orc_gut_project(const GutConfig* cfg, const GrutCamera* cam, const real* pose_start7, const real* pose_end7,
                    uint32_t N, int n_active_features, const real* density12, const real* sph,
                    uint32_t* tiles_count, real* proj_pos, real* conic_opacity, real* extent, real* depth, real* rgb,
                    int32_t* visibility) {
    const orc_frame_poses fp = frame_poses(pose_start7, pose_end7);
    const int gx = tile_grid_dim(cam->width), gy = tile_grid_dim(cam->height);
    const int ncoef = (cfg->particle_radiance_sph_degree + 1) * (cfg->particle_radiance_sph_degree + 1);
    const real alpha_ut = (real)cfg->ut_alpha, beta_ut = (real)cfg->ut_beta, kappa_ut = (real)cfg->ut_kappa;
    const real D = 3;
    const real lambda = alpha_ut * alpha_ut * (D + kappa_ut) - D;       /* :150 */
    const real delta_ut = r_sqrt(alpha_ut * alpha_ut * (D + kappa_ut)); /* setup_3dgut.py:44 */
    const real w0m = lambda / (D + lambda), wi = 1 / (2 * (D + lambda)); /* :161,163 */
    const real w0c = lambda / (D + lambda) + (1 - alpha_ut * alpha_ut + beta_ut); /* :201 */
    const real margin = (real)cfg->ut_in_image_margin_factor;

#pragma omp parallel for schedule(static, 256)   /* particles are independent */
    for (uint32_t i = 0; i < N; ++i) {
        const real* pd = density12 + 12 * (size_t)i;
        const v3 pos = v3_make(pd[0], pd[1], pd[2]);
        const real opacity = pd[3];
        const v4 q = {pd[4], pd[5], pd[6], pd[7]};
        const v3 scl = v3_make(pd[8], pd[9], pd[10]);
        tiles_count[i] = 0; visibility[i] = 0;
        proj_pos[2 * i] = proj_pos[2 * i + 1] = 0;
        conic_opacity[4 * i] = conic_opacity[4 * i + 1] = conic_opacity[4 * i + 2] = conic_opacity[4 * i + 3] = 0;
        extent[2 * i] = extent[2 * i + 1] = 0; depth[i] = 0;
        rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = 0;

        /* :131-140 */
        if (opacity < (real)cfg->particle_kernel_min_alpha) continue;
        const real view_z = v3_dot(fp.view_R.r[2], pos) + fp.view_t.z;
        if (view_z < R_(0.2)) continue; /* threedgut.cuh:57 ParticleMinSensorZ */

        const m33 rotT = quat_wxyz_to_rotT(q); /* rotation(params)[i] == rotT.r[i] == R[:,i] */
        real sp[7][2];
        int nvalid = 0;
        nvalid += project_point_with_shutter(cam, fp.start, fp.end, cfg->n_rolling_shutter_iterations, pos, margin, &sp[0][0], &sp[0][1]);
        real cx = sp[0][0] * w0m, cy = sp[0][1] * w0m;
        const real sarr[3] = {scl.x, scl.y, scl.z};
        for (int k = 0; k < 3; ++k) {
            const v3 d = v3_scale(rotT.r[k], delta_ut * sarr[k]);
            nvalid += project_point_with_shutter(cam, fp.start, fp.end, cfg->n_rolling_shutter_iterations, v3_add(pos, d), margin, &sp[k + 1][0], &sp[k + 1][1]);
            cx += wi * sp[k + 1][0]; cy += wi * sp[k + 1][1];
            nvalid += project_point_with_shutter(cam, fp.start, fp.end, cfg->n_rolling_shutter_iterations, v3_sub(pos, d), margin, &sp[k + 4][0], &sp[k + 4][1]);
            cx += wi * sp[k + 4][0]; cy += wi * sp[k + 4][1];
        }
        if (cfg->ut_require_all_sigma_points_valid ? (nvalid < 7) : (nvalid == 0)) continue;
        real cov[3];
        {
            const real dx = sp[0][0] - cx, dy = sp[0][1] - cy;
            cov[0] = w0c * dx * dx; cov[1] = w0c * dx * dy; cov[2] = w0c * dy * dy;
        }
        for (int k = 1; k < 7; ++k) {
            const real dx = sp[k][0] - cx, dy = sp[k][1] - cy;
            cov[0] += wi * dx * dx; cov[1] += wi * dx * dy; cov[2] += wi * dy * dy;
        }
        real ext[2], co[4], pmax;
        if (!extent_conic_opacity(cfg, cov, opacity, ext, co, &pmax)) continue;
        visibility[i] = 1;

        /* :279-293 */
        const tile_bbox bb = tile_space_bbox(gx, gy, cx, cy, ext[0], ext[1]);
        uint32_t ntiles = 0;
        if (cfg->tile_based_culling) {
            for (int y = bb.miny; y < bb.maxy; ++y)
                for (int x = bb.minx; x < bb.maxx; ++x)
                    if (tile_min_power((real)x, (real)y, co, cx, cy) < pmax) ntiles++;
        } else {
            ntiles = (uint32_t)((bb.maxx - bb.minx) * (bb.maxy - bb.miny));
        }
        tiles_count[i] = ntiles;
        if (ntiles == 0) continue;

        /* :304-321 */
        const v3 ray = v3_sub(pos, fp.sensor_world_pos);
        const real dist = r_sqrt(v3_dot(ray, ray));
        const v3 dir = v3_scale(ray, 1 / dist);
        const v3 c = sh_radiance_unclamped(n_active_features, sph + (size_t)i * 3 * ncoef, dir);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
        proj_pos[2 * i] = cx; proj_pos[2 * i + 1] = cy;
        for (int k = 0; k < 4; ++k) conic_opacity[4 * i + k] = co[k];
        extent[2 * i] = ext[0]; extent[2 * i + 1] = ext[1];
        depth[i] = cfg->global_z_order ? view_z : dist;
    }
    return 0;
}

/* --------------------------------------------------------------------------------------
 * binning: scan + expand + sort + ranges — gutRenderer.cu:302-372, gutProjector.cuh:324-388
 * ------------------------------------------------------------------------------------ */
/* gutRenderer.cu:79-94 higherMsb */
uint32_t orc_higher_msb(uint32_t n) {
    uint32_t msb = 16, step = 16;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

typedef struct { uint64_t key; uint32_t val; uint32_t seq; } orc_kv;
static int orc_kv_cmp(const void* a, const void* b) {
    const orc_kv* x = (const orc_kv*)a; const orc_kv* y = (const orc_kv*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0); /* stable == LSD radix sort */
}

static uint32_t depth_key_bits(real d) { float f = (float)d; uint32_t u; memcpy(&u, &f, 4); return u; }

/* Returns the number of tile intersections I.  If sorted_keys==NULL only counts.
 * sorted_keys[I] u64, sorted_idx[I] u32, tile_ranges[tiles*2] u32 (zero-filled here). */
uint64_t orc_gut_bin(const GutConfig* cfg, int width, int height, uint32_t N,
                     const uint32_t* tiles_count, const real* proj_pos, const real* conic_opacity,
                     const real* extent, const real* depth,
                     uint64_t* sorted_keys, uint32_t* sorted_idx, uint32_t* tile_ranges) {
    const int gx = tile_grid_dim(width), gy = tile_grid_dim(height);
    uint64_t total = 0;
    for (uint32_t i = 0; i < N; ++i) total += tiles_count[i];
    if (!sorted_keys || total == 0) return total;

    orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * total);
    uint64_t off = 0;
    for (uint32_t i = 0; i < N; ++i) {
        const uint64_t max_off = off + tiles_count[i];
        const real ex = extent[2 * i], ey = extent[2 * i + 1];
        if (ex <= R_(1e-06)) { /* expand :346-349; count is 0 for such particles */
            for (; off < max_off; ++off) { kv[off].key = ~0ull; kv[off].val = ORC_INVALID_IDX; kv[off].seq = (uint32_t)off; }
            continue;
        }
        const uint32_t dk = depth_key_bits(depth[i]);
        const real cx = proj_pos[2 * i], cy = proj_pos[2 * i + 1];
        const tile_bbox bb = tile_space_bbox(gx, gy, cx, cy, ex, ey);
        if (cfg->tile_based_culling) {
            const real* co = conic_opacity + 4 * (size_t)i;
            const real pmax = r_log(co[3] / (real)cfg->particle_kernel_min_alpha);
            for (int y = bb.miny; y < bb.maxy && off < max_off; ++y)
                for (int x = bb.minx; x < bb.maxx && off < max_off; ++x)
                    if (tile_min_power((real)x, (real)y, co, cx, cy) < pmax) {
                        kv[off].key = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dk;
                        kv[off].val = i; kv[off].seq = (uint32_t)off; off++;
                    }
            for (; off < max_off; ++off) { /* :372-376 pad */
                kv[off].key = ((uint64_t)0xFFFFFFFFu << 32) | 0x7F7FFFFFu; kv[off].val = ORC_INVALID_IDX; kv[off].seq = (uint32_t)off;
            }
        } else {
            for (int y = bb.miny; y < bb.maxy; ++y)
                for (int x = bb.minx; x < bb.maxx; ++x) {
                    kv[off].key = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dk;
                    kv[off].val = i; kv[off].seq = (uint32_t)off; off++;
                }
        }
    }
    /* cub::DeviceRadixSort::SortPairs on bits [0, 32+higherMsb(tiles)) : stable sort of the masked key */
    const uint32_t bits = 32 + orc_higher_msb((uint32_t)(gx * gy));
    const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    uint64_t* full = (uint64_t*)malloc(sizeof(uint64_t) * total);
    for (uint64_t k = 0; k < total; ++k) { full[k] = kv[k].key; kv[k].key &= mask; }
    qsort(kv, total, sizeof(orc_kv), orc_kv_cmp);
    for (uint64_t k = 0; k < total; ++k) { sorted_keys[k] = full[kv[k].seq]; sorted_idx[k] = kv[k].val; }
    free(full); free(kv);

    /* computeSortedTileRangeIndices, gutRenderer.cu:46-76 */
    memset(tile_ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    for (uint64_t k = 0; k < total; ++k) {
        const uint32_t t = (uint32_t)(sorted_keys[k] >> 32);
        const int valid = t != 0xFFFFFFFFu;
        if (k == 0) {
            if (valid) tile_ranges[2 * t] = 0;
        } else {
            const uint32_t pt = (uint32_t)(sorted_keys[k - 1] >> 32);
            if (pt != t) {
                if (pt != 0xFFFFFFFFu) tile_ranges[2 * pt + 1] = (uint32_t)k;
                if (valid) tile_ranges[2 * t] = (uint32_t)k;
            }
        }
        if (valid && k == total - 1) tile_ranges[2 * t + 1] = (uint32_t)total;
    }
    return total;
}

/* --------------------------------------------------------------------------------------
 * per-hit math
 * ------------------------------------------------------------------------------------ */
typedef struct {
    v3 pos, scl; v4 quat; m33 rotT; real density;
} orc_particle;

static orc_particle load_particle(const real* pd) {
    orc_particle p;
    p.pos = v3_make(pd[0], pd[1], pd[2]); p.density = pd[3];
    p.quat.x = pd[4]; p.quat.y = pd[5]; p.quat.z = pd[6]; p.quat.w = pd[7];
    p.scl = v3_make(pd[8], pd[9], pd[10]);
    p.rotT = quat_wxyz_to_rotT(p.quat);
    return p;
}

/* gaussianParticles.slang:207-242 hit(): returns accept; alpha, hitT(depth) */
static int density_hit_ex(const GutConfig* cfg, v3 ro, v3 rd, const orc_particle* p, real* alpha, real* hitT, real* resp_out);
static int density_hit(const GutConfig* cfg, v3 ro, v3 rd, const orc_particle* p, real* alpha, real* hitT) {
    real resp;
    return density_hit_ex(cfg, ro, rd, p, alpha, hitT, &resp);
}
static int density_hit_ex(const GutConfig* cfg, v3 ro, v3 rd, const orc_particle* p, real* alpha, real* hitT, real* resp_out) {
    const v3 giscl = v3_make(1 / p->scl.x, 1 / p->scl.y, 1 / p->scl.z);
    const v3 gposc = v3_sub(ro, p->pos);
    const v3 gposcr = v3_mul_rows(gposc, &p->rotT);
    const v3 gro = v3_mul(giscl, gposcr);
    const v3 rdr = v3_mul_rows(rd, &p->rotT);
    const v3 grdu = v3_mul(giscl, rdr);
    const v3 grd = v3_scale(grdu, 1 / r_sqrt(v3_dot(grdu, grdu))); /* slang normalize() :109 */
    const v3 gcrod = v3_cross(grd, gro);
    const real gray = v3_dot(gcrod, gcrod);
    const real resp = particle_response(cfg->particle_kernel_degree, gray);
    *resp_out = resp;
    *alpha = r_min((real)cfg->particle_kernel_max_alpha, resp * p->density);
    const int accept = (resp > (real)cfg->particle_kernel_min_response) && (*alpha > (real)cfg->particle_kernel_min_alpha);
    if (accept) { /* :181-190 canonicalRayIntersection */
        const v3 cg = v3_scale(grd, v3_dot(grd, v3_scale(gro, -1)));
        const v3 grds = v3_mul(p->scl, cg);
        *hitT = r_sqrt(v3_dot(grds, grds));
    }
    return accept;
}

/* Exported for known-answer tests against the reference header compiled on the host (oracle/_ref):
 * threedgut::processHitFwd<degree,false,false>, models/gaussianParticles.cuh:350-422.
 * state = {T, rgb[3], depth}; feat = per-particle radiance (already clamped). returns accept. */
int orc_gut_process_hit_fwd(int degree, real min_response, real min_alpha, real max_alpha,
                            const real* ray_o, const real* ray_d, const real* density12, const real* feat3,
                            real* T, real* rgb, real* depth) {
    GutConfig cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.particle_kernel_degree = degree; cfg.particle_kernel_min_response = (float)min_response;
    cfg.particle_kernel_min_alpha = (float)min_alpha; cfg.particle_kernel_max_alpha = (float)max_alpha;
    const orc_particle p = load_particle(density12);
    real alpha, hitT;
    /* NB thresholds pass through float in GutConfig; callers use float-representable values */
    const int acc = density_hit(&cfg, v3_make(ray_o[0], ray_o[1], ray_o[2]), v3_make(ray_d[0], ray_d[1], ray_d[2]), &p, &alpha, &hitT);
    if (acc) {
        const real w = alpha * (*T);
        rgb[0] += w * feat3[0]; rgb[1] += w * feat3[1]; rgb[2] += w * feat3[2];
        *T *= (1 - alpha);
        *depth += hitT * w;
    }
    return acc;
}

typedef struct {
    real T;          /* running transmittance */
    v3 feat;         /* running radiance */
    real hitT;       /* running depth */
    real T_fin; v3 feat_fin; real hitT_fin; /* forward results ("Backward" fields, rayPayloadBackward.cuh:21-28) */
    real T_grad; v3 feat_grad; real hitT_grad;
} orc_bwd_ray;

/* threedgut::processHitBwd<degree,false,false>, models/gaussianParticles.cuh:484-751.
 * Writes this hit's gradient into g_density12[12] (pos, density, quat wxyz, scale, pad) and g_feat[3]. */
static void process_hit_bwd(const GutConfig* cfg, v3 ro, v3 rd, const orc_particle* p, v3 feat /* clamped radiance */,
                            orc_bwd_ray* ray, real* g_density12, real* g_feat) {
    const v3 gscl = p->scl;
    const v3 giscl = v3_make(1 / gscl.x, 1 / gscl.y, 1 / gscl.z);
    const v3 gposc = v3_sub(ro, p->pos);
    const v3 gposcr = v3_mul_rows(gposc, &p->rotT);
    const v3 gro = v3_mul(giscl, gposcr);
    const v3 rdr = v3_mul_rows(rd, &p->rotT);
    const v3 grdu = v3_mul(giscl, rdr);
    const v3 grd = v3_safe_normalize(grdu);
    const v3 gcrod = v3_cross(grd, gro);
    const real gray = v3_dot(gcrod, gcrod);
    const real gres = particle_response(cfg->particle_kernel_degree, gray);
    const real galpha = r_min((real)cfg->particle_kernel_max_alpha, gres * p->density);
    if (!((gres > (real)cfg->particle_kernel_min_response) && (galpha > (real)cfg->particle_kernel_min_alpha))) return;

    const real minT = (real)cfg->min_transmittance;
    const real pdot = v3_dot(grd, v3_scale(gro, -1));
    const v3 grdd = v3_scale(grd, pdot);
    const v3 grds = v3_mul(gscl, grdd);
    const real gsq = v3_dot(grds, grds);
    const real gdist = r_sqrt(gsq);
    const real T = ray->T;
    const real weight = galpha * T;
    const real nextT = (1 - galpha) * T;

    ray->hitT += weight * gdist;
    const real resHitT = r_max(nextT <= minT ? 0 : (ray->hitT_fin - ray->hitT) / nextT, 0);
    const real galphaRayHitGrd = (gdist - resHitT) * T * ray->hitT_grad;
    const v3 grdsRayHitGrd = gsq > 0 ? v3_scale(grds, (2 * weight) / (2 * gdist) * ray->hitT_grad) : v3_make(0, 0, 0);
    const v3 gsclRayHitGrd = v3_mul(grdd, grdsRayHitGrd);
    const real grdScaledDot = v3_dot(v3_mul(grdsRayHitGrd, gscl), grd);
    const v3 grdRayHitGrd = v3_sub(v3_scale(v3_mul(gscl, grdsRayHitGrd), pdot), v3_scale(gro, grdScaledDot));
    const v3 groRayHitGrd = v3_scale(grd, -grdScaledDot);

    const real resTrm = galpha < R_(0.999999) ? ray->T_fin / (1 - galpha) : T;
    const real galphaRayDnsGrd = resTrm * -ray->T_grad;

    /* PerRayRadiance=false: grad = particle radiance, radiance grad = rayRadGrad*weight (:601-606) */
    g_feat[0] = ray->feat_grad.x * weight; g_feat[1] = ray->feat_grad.y * weight; g_feat[2] = ray->feat_grad.z * weight;
    ray->feat = v3_add(ray->feat, v3_scale(feat, weight));
    v3 resRad;
    if (nextT <= minT) resRad = v3_make(0, 0, 0);
    else {
        resRad = v3_scale(v3_sub(ray->feat_fin, ray->feat), 1 / nextT);
        resRad = v3_make(r_max(resRad.x, 0), r_max(resRad.y, 0), r_max(resRad.z, 0));
    }
    const real common = galphaRayHitGrd + galphaRayDnsGrd + T * (feat.x - resRad.x) * ray->feat_grad.x +
                        T * (feat.y - resRad.y) * ray->feat_grad.y + T * (feat.z - resRad.z) * ray->feat_grad.z;
    g_density12[3] = gres * common;
    const real gresGrd = p->density * common;
    const real grayGrd = particle_response_grd(cfg->particle_kernel_degree, gray, gres, gresGrd);

    const v3 gcrodGrd = v3_scale(gcrod, 2 * grayGrd);
    const v3 grdGrd = v3_make(gcrodGrd.z * gro.y - gcrodGrd.y * gro.z, gcrodGrd.x * gro.z - gcrodGrd.z * gro.x, gcrodGrd.y * gro.x - gcrodGrd.x * gro.y);
    const v3 groGrd = v3_make(gcrodGrd.y * grd.z - gcrodGrd.z * grd.y, gcrodGrd.z * grd.x - gcrodGrd.x * grd.z, gcrodGrd.x * grd.y - gcrodGrd.y * grd.x);

    const v3 groTot = v3_add(groGrd, groRayHitGrd);
    const v3 gsclGrdGro = v3_mul(v3_make(-gposcr.x / (gscl.x * gscl.x), -gposcr.y / (gscl.y * gscl.y), -gposcr.z / (gscl.z * gscl.z)), groTot);
    const v3 gposcrGrd = v3_mul(giscl, groTot);
    const v3 gposcGrd = matmul_bw_vec(&p->rotT, gposcrGrd);
    const v4 grotGrdPoscr = matmul_bw_quat(gposc, gposcrGrd, p->quat);
    g_density12[0] = -gposcGrd.x; g_density12[1] = -gposcGrd.y; g_density12[2] = -gposcGrd.z;

    const v3 grduGrd = v3_safe_normalize_bw(grdu, v3_add(grdGrd, grdRayHitGrd));
    const v3 sclGrd = v3_add(v3_add(gsclRayHitGrd, gsclGrdGro),
                             v3_mul(v3_make(-rdr.x / (gscl.x * gscl.x), -rdr.y / (gscl.y * gscl.y), -rdr.z / (gscl.z * gscl.z)), grduGrd));
    g_density12[8] = sclGrd.x; g_density12[9] = sclGrd.y; g_density12[10] = sclGrd.z;
    const v3 rdrGrd = v3_mul(giscl, grduGrd);
    const v4 grotGrdRd = matmul_bw_quat(rd, rdrGrd, p->quat);
    g_density12[4] = grotGrdPoscr.x + grotGrdRd.x; g_density12[5] = grotGrdPoscr.y + grotGrdRd.y;
    g_density12[6] = grotGrdPoscr.z + grotGrdRd.z; g_density12[7] = grotGrdPoscr.w + grotGrdRd.w;

    ray->T = nextT;
}

/* exported KAT wrapper. ray_state: in/out {T, feat[3], hitT}; fin {T_fin, feat_fin[3], hitT_fin}; grads {T_grad, feat_grad[3], hitT_grad}
 * where T_grad is the gradient w.r.t. transmittance (= -d/d opacity). Outputs the per-hit gradient. */
void orc_gut_process_hit_bwd(int degree, real min_response, real min_alpha, real max_alpha, real min_transmittance,
                             const real* ray_o, const real* ray_d, const real* density12, const real* feat3,
                             real* state5, const real* fin5, const real* grads5, real* g_density12, real* g_feat3) {
    GutConfig cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.particle_kernel_degree = degree; cfg.particle_kernel_min_response = (float)min_response;
    cfg.particle_kernel_min_alpha = (float)min_alpha; cfg.particle_kernel_max_alpha = (float)max_alpha;
    cfg.min_transmittance = (float)min_transmittance;
    const orc_particle p = load_particle(density12);
    orc_bwd_ray r;
    r.T = state5[0]; r.feat = v3_make(state5[1], state5[2], state5[3]); r.hitT = state5[4];
    r.T_fin = fin5[0]; r.feat_fin = v3_make(fin5[1], fin5[2], fin5[3]); r.hitT_fin = fin5[4];
    r.T_grad = grads5[0]; r.feat_grad = v3_make(grads5[1], grads5[2], grads5[3]); r.hitT_grad = grads5[4];
    for (int k = 0; k < 12; ++k) g_density12[k] = 0;
    g_feat3[0] = g_feat3[1] = g_feat3[2] = 0;
    process_hit_bwd(&cfg, v3_make(ray_o[0], ray_o[1], ray_o[2]), v3_make(ray_d[0], ray_d[1], ray_d[2]), &p,
                    v3_make(feat3[0], feat3[1], feat3[2]), &r, g_density12, g_feat3);
    state5[0] = r.T; state5[1] = r.feat.x; state5[2] = r.feat.y; state5[3] = r.feat.z; state5[4] = r.hitT;
}

/* --------------------------------------------------------------------------------------
 * K > 0 ("sorted") backward per hit: processHitParticle<Backward> (gutKBufferRenderer.cuh:158-198) =
 * particleFeaturesIntegrateBwd (shRadiativeParticles.slang:210-256) + particleDensityProcessHitBwdToBuffer
 * (gaussianParticles.slang:420-479): Slang reverse-mode of the BACK-TO-FRONT lerp form, the ray state being un-blended
 * by 1/(1-alpha) as the hits are visited front to back.  Differences to the K = 0 hand-written chain: no clamping of the
 * residuals, and the gradient does not pass through alpha = min(MaxParticleAlpha, .) when the clamp is active.
 * state: Tb, Cb, Db = "behind" values (start at the forward results); gT, gC, gD = running upstream gradients. */
typedef struct {
    real Tb; v3 Cb; real Db;
    real gT; v3 gC; real gD;
} orc_kbwd_ray;

static void process_hit_bwd_k(const GutConfig* cfg, v3 ro, v3 rd, const orc_particle* p, v3 feat, real alpha, real hitT,
                              orc_kbwd_ray* r, real* g_density12, real* g_feat) {
    for (int k = 0; k < 12; ++k) g_density12[k] = 0;
    g_feat[0] = g_feat[1] = g_feat[2] = 0;
    if (!(alpha > 0)) return;
    const real w = 1 / (1 - alpha);
    /* features: I_front = lerp(I_behind, f, alpha) */
    r->Cb = v3_scale(v3_sub(r->Cb, v3_scale(feat, alpha)), w);
    real dalpha = (feat.x - r->Cb.x) * r->gC.x + (feat.y - r->Cb.y) * r->gC.y + (feat.z - r->Cb.z) * r->gC.z;
    g_feat[0] = alpha * r->gC.x; g_feat[1] = alpha * r->gC.y; g_feat[2] = alpha * r->gC.z;
    r->gC = v3_scale(r->gC, 1 - alpha);
    /* density: T_out = T_in (1 - alpha), D_front = lerp(D_behind, depth, alpha) */
    r->Tb *= w;
    r->Db = (r->Db - hitT * alpha) * w;
    dalpha += (hitT - r->Db) * r->gD - r->Tb * r->gT;
    const real ddepth = alpha * r->gD;
    r->gD *= (1 - alpha);
    r->gT *= (1 - alpha);

    /* hit(): recompute the canonical ray (gaussianParticles.slang:96-110, 207-242) */
    const v3 gscl = p->scl;
    const v3 giscl = v3_make(1 / gscl.x, 1 / gscl.y, 1 / gscl.z);
    const v3 gposc = v3_sub(ro, p->pos);
    const v3 gposcr = v3_mul_rows(gposc, &p->rotT);
    const v3 gro = v3_mul(giscl, gposcr);
    const v3 rdr = v3_mul_rows(rd, &p->rotT);
    const v3 grdu = v3_mul(giscl, rdr);
    const v3 grd = v3_scale(grdu, 1 / r_sqrt(v3_dot(grdu, grdu)));
    const v3 gcrod = v3_cross(grd, gro);
    const real gray = v3_dot(gcrod, gcrod);
    const real gres = particle_response(cfg->particle_kernel_degree, gray);
    /* alpha = min(MaxAlpha, gres * density): reverse-mode of min passes the gradient to the smaller argument */
    real dres = 0, ddens = 0;
    if (gres * p->density < (real)cfg->particle_kernel_max_alpha) { dres = p->density * dalpha; ddens = gres * dalpha; }
    g_density12[3] = ddens;
    const real grayGrd = particle_response_grd(cfg->particle_kernel_degree, gray, gres, dres);
    /* depth = |scale * grd * dot(grd, -gro)| */
    const real pdot = v3_dot(grd, v3_scale(gro, -1));
    const v3 grdd = v3_scale(grd, pdot);
    const v3 grds = v3_mul(gscl, grdd);
    const real gsq = v3_dot(grds, grds);
    const real gdist = r_sqrt(gsq);
    const v3 grdsGrd = gsq > 0 ? v3_scale(grds, ddepth / gdist) : v3_make(0, 0, 0);
    const v3 gsclHit = v3_mul(grdd, grdsGrd);
    const real sdot = v3_dot(v3_mul(grdsGrd, gscl), grd);
    const v3 grdHit = v3_sub(v3_scale(v3_mul(gscl, grdsGrd), pdot), v3_scale(gro, sdot));
    const v3 groHit = v3_scale(grd, -sdot);
    /* grayDist = |cross(grd, gro)|^2 */
    const v3 gcrodGrd = v3_scale(gcrod, 2 * grayGrd);
    const v3 grdGrd = v3_make(gcrodGrd.z * gro.y - gcrodGrd.y * gro.z, gcrodGrd.x * gro.z - gcrodGrd.z * gro.x, gcrodGrd.y * gro.x - gcrodGrd.x * gro.y);
    const v3 groGrd = v3_make(gcrodGrd.y * grd.z - gcrodGrd.z * grd.y, gcrodGrd.z * grd.x - gcrodGrd.x * grd.z, gcrodGrd.x * grd.y - gcrodGrd.y * grd.x);
    const v3 groTot = v3_add(groGrd, groHit);
    const v3 gsclGro = v3_mul(v3_make(-gposcr.x / (gscl.x * gscl.x), -gposcr.y / (gscl.y * gscl.y), -gposcr.z / (gscl.z * gscl.z)), groTot);
    const v3 gposcrGrd = v3_mul(giscl, groTot);
    const v3 gposcGrd = matmul_bw_vec(&p->rotT, gposcrGrd);
    const v4 gq1 = matmul_bw_quat(gposc, gposcrGrd, p->quat);
    g_density12[0] = -gposcGrd.x; g_density12[1] = -gposcGrd.y; g_density12[2] = -gposcGrd.z;
    const v3 grduGrd = v3_safe_normalize_bw(grdu, v3_add(grdGrd, grdHit));
    const v3 sclGrd = v3_add(v3_add(gsclHit, gsclGro),
                             v3_mul(v3_make(-rdr.x / (gscl.x * gscl.x), -rdr.y / (gscl.y * gscl.y), -rdr.z / (gscl.z * gscl.z)), grduGrd));
    g_density12[8] = sclGrd.x; g_density12[9] = sclGrd.y; g_density12[10] = sclGrd.z;
    const v4 gq2 = matmul_bw_quat(rd, v3_mul(giscl, grduGrd), p->quat);
    g_density12[4] = gq1.x + gq2.x; g_density12[5] = gq1.y + gq2.y; g_density12[6] = gq1.z + gq2.z; g_density12[7] = gq1.w + gq2.w;
}

/* SH KATs: models/gaussianParticles.cuh:68-100 (radianceFromSpH) */
void orc_sh_radiance(int deg, int max_deg, const real* coeffs, const real* dir3, int clamped, real* out3) {
    (void)max_deg;
    v3 c = sh_radiance_unclamped(deg, coeffs, v3_make(dir3[0], dir3[1], dir3[2]));
    if (clamped) { c.x = r_max(c.x, 0); c.y = r_max(c.y, 0); c.z = r_max(c.z, 0); }
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}

/* --------------------------------------------------------------------------------------
 * rays — rayPayload.cuh:75-108, bounding_box.h:89-140
 * ------------------------------------------------------------------------------------ */
static void aabb_ray_intersect(real lo, real hi, v3 o, v3 d, real* tmin_o, real* tmax_o) {
    const real big = R_(3.4028234663852886e+38);
    real tmin = (lo - o.x) / d.x, tmax = (hi - o.x) / d.x;
    if (tmin > tmax) { real t = tmin; tmin = tmax; tmax = t; }
    real tymin = (lo - o.y) / d.y, tymax = (hi - o.y) / d.y;
    if (tymin > tymax) { real t = tymin; tymin = tymax; tymax = t; }
    if (tmin > tymax || tymin > tmax) { *tmin_o = big; *tmax_o = big; return; }
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    real tzmin = (lo - o.z) / d.z, tzmax = (hi - o.z) / d.z;
    if (tzmin > tzmax) { real t = tzmin; tzmin = tzmax; tzmax = t; }
    if (tmin > tzmax || tzmin > tmax) { *tmin_o = big; *tmax_o = big; return; }
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    *tmin_o = tmin; *tmax_o = tmax;
}

typedef struct { v3 o, d; real tmin, tmax; int valid; } orc_ray;
static orc_ray init_ray(const orc_frame_poses* fp, const real* ro, const real* rd) {
    orc_ray r;
    r.o = v3_add(m33_apply(&fp->s2w_R, v3_make(ro[0], ro[1], ro[2])), fp->s2w_t);
    r.d = m33_apply(&fp->s2w_R, v3_make(rd[0], rd[1], rd[2]));
    aabb_ray_intersect(R_(-1e6), R_(1e6), r.o, r.d, &r.tmin, &r.tmax); /* splatRaster.cpp:240 */
    r.tmin = r_max(r.tmin, 0);
    r.valid = r.tmax > r.tmin;
    return r;
}

/* --------------------------------------------------------------------------------------
 * render forward — gutKBufferRenderer.cuh:228-352 (+ k-buffer :62-122), rayPayload.cuh:160-193
 * ------------------------------------------------------------------------------------ */
#define ORC_MAX_K 64
typedef struct { uint32_t idx; real hitT, alpha; } orc_hit;

static void integrate_hit(const GutConfig* cfg, const orc_hit* h, const real* rgb, real* T, v3* C, real* D, uint32_t* cnt, int* alive) {
    /* processHitParticle fwd :199-225; gaussianParticles.slang:244-274; shRadiativeParticles.slang:83-99 */
    const real w = h->alpha * (*T);
    *D += h->hitT * w;
    *T *= (1 - h->alpha);
    if (w > 0) {
        const real* c = rgb + 3 * (size_t)h->idx;
        C->x += r_max(c[0], 0) * w; C->y += r_max(c[1], 0) * w; C->z += r_max(c[2], 0) * w;
        (*cnt)++;
    }
    if (*T < (real)cfg->min_transmittance) *alive = 0;
}

int orc_gut_render_fwd(const GutConfig* cfg, int width, int height, const real* pose_start7, const real* pose_end7,
                       const real* density12, const real* rgb, const uint32_t* sorted_idx, const uint32_t* tile_ranges,
                       const real* ray_o, const real* ray_d, real* out_fd, real* out_dist, real* out_cnt) {
    const orc_frame_poses fp = frame_poses(pose_start7, pose_end7);
    const int gx = tile_grid_dim(width);
    const int K = cfg->k_buffer_size;
    if (K > ORC_MAX_K) return -1;
#pragma omp parallel for schedule(dynamic, 16)
    for (int pix = 0; pix < width * height; ++pix) {
        const int x = pix % width, y = pix / width;
        const orc_ray ray = init_ray(&fp, ray_o + 3 * (size_t)pix, ray_d + 3 * (size_t)pix);
        if (!ray.valid) continue; /* finalizeRay returns early: outputs keep their initial values */
        const uint32_t tile = (uint32_t)((y / ORC_TILE) * gx + (x / ORC_TILE));
        const uint32_t beg = tile_ranges[2 * tile], end = tile_ranges[2 * tile + 1];
        real T = 1, D = 0; v3 C = v3_make(0, 0, 0); uint32_t cnt = 0; int alive = 1;
        orc_hit kbuf[ORC_MAX_K]; int nhits = 0;
        for (int k = 0; k < K; ++k) { kbuf[k].idx = ORC_INVALID_IDX; kbuf[k].hitT = -1; kbuf[k].alpha = 0; }
        for (uint32_t e = beg; e < end && alive; ++e) {
            const uint32_t idx = sorted_idx[e];
            if (idx == ORC_INVALID_IDX) break;
            const orc_particle p = load_particle(density12 + 12 * (size_t)idx);
            orc_hit h; h.idx = idx; h.hitT = -1; h.alpha = 0;
            if (density_hit(cfg, ray.o, ray.d, &p, &h.alpha, &h.hitT) && h.hitT > ray.tmin && h.hitT < ray.tmax) {
                if (K == 0) {
                    integrate_hit(cfg, &h, rgb, &T, &C, &D, &cnt, &alive);
                } else {
                    /* :331-338 ; HitParticleKBufferT::insert :76-91 */
                    const int full = nhits == K;
                    if (full) {
                        integrate_hit(cfg, &kbuf[0], rgb, &T, &C, &D, &cnt, &alive);
                        kbuf[0].hitT = -1;
                    } else nhits++;
                    for (int i = K - 1; i >= 0; --i)
                        if (h.hitT > kbuf[i].hitT) { const orc_hit t = kbuf[i]; kbuf[i] = h; h = t; }
                }
            }
        }
        if (K > 0) /* :343-351 */
            for (int i = 0; alive && i < nhits; ++i) integrate_hit(cfg, &kbuf[K - nhits + i], rgb, &T, &C, &D, &cnt, &alive);
        out_fd[4 * (size_t)pix] = C.x; out_fd[4 * (size_t)pix + 1] = C.y; out_fd[4 * (size_t)pix + 2] = C.z;
        out_fd[4 * (size_t)pix + 3] = 1 - T;
        out_dist[pix] = D;
        if (cfg->enable_hitcounts) out_cnt[pix] = (real)cnt;
    }
    return 0;
}

/* --------------------------------------------------------------------------------------
 * render forward with NEURAL HARMONIC FEATURES (model.feature_type = nht, FEATURE_TRANSFORM_TYPE 1), K = 0:
 * gutKBufferRenderer.cuh:199-225 processHitParticle (featureIntegrateFwd of featuresFromBuffer at the hit's canonical intersection),
 * :228-352 the tile loop; canonical intersection gaussianParticles.slang:181-190; the feature model
 * neuralHarmonicFeaturesParticle.slang:85-97 (fetch), :117-127 (barycentric weights in the canonical tetrahedron :47-66),
 * :146-196 (blend + activation), :198-211 (integration: += features * weight if weight > 0); write-out of
 * RAY_FEATURE_DIM + 1 channels rayPayload.cuh:160-193.  Pinned by tests/golden/gut_nht.npz (the reference's renderer around a
 * restatement of the .slang feature model, oracle/ref/shim/threedgutSlang.cuh).
 * nht = {particle_feature_dim K, interp_point_dim, support (0 centre, 1 tetrahedra), activation (0 none, 1 siren, 2 sincos, 3 relu),
 *        num_frequencies}; features [N, K]; out_fd [H, W, ray_dim + 1]. */
#define ORC_NHT_MAX_RAY_DIM 64
static int nht_ray_dim(const int* nht) {
    return nht[3] == 2 ? nht[1] * nht[4] * 2 : ((nht[3] == 0 || nht[3] == 3) ? nht[1] : nht[1] * nht[4]);
}
static void nht_features_at(const int* nht, const real* row, v3 P, real* out) {
    const int ipd = nht[1], act = nht[3], nf = nht[4];
    real base[ORC_NHT_MAX_RAY_DIM];
    for (int n = 0; n < ipd; ++n) base[n] = row[n];
    if (nht[2] == 1) {
        const real edge = R_(4.898979485566356), face_h = R_(4.242640687119285), height = R_(4.0), face_in = R_(1.4142135623730951), in_r = R_(1.0);
        const v3 v0 = v3_make(R_(0.5) * edge, -face_in, R_(-1.0)), v1 = v3_make(R_(-0.5) * edge, -face_in, R_(-1.0));
        const v3 v2 = v3_make(0, face_h - face_in, R_(-1.0)), v3_ = v3_make(0, 0, height - in_r);
        const v3 e1 = v3_sub(v1, v0), e2 = v3_sub(v2, v0), e3 = v3_sub(v3_, v0);
        const v3 c23 = v3_cross(e2, e3);
        const real inv_det = 1 / v3_dot(e1, c23);
        const v3 d = v3_sub(P, v0);
        real w[4];
        w[1] = v3_dot(d, c23) * inv_det;
        w[2] = v3_dot(e1, v3_cross(d, e3)) * inv_det;
        w[3] = v3_dot(e1, v3_cross(e2, d)) * inv_det;
        w[0] = 1 - w[1] - w[2] - w[3];
        for (int n = 0; n < ipd; ++n) base[n] *= w[0];
        for (int k = 1; k < 4; ++k)
            for (int n = 0; n < ipd; ++n) base[n] += w[k] * row[k * ipd + n];
    }
    if (act == 0) { for (int i = 0; i < ipd; ++i) out[i] = base[i]; }
    else if (act == 3) { for (int i = 0; i < ipd; ++i) out[i] = r_max(0, base[i]); }
    else if (act == 2) {
        for (int k = 0; k < ipd; ++k)
            for (int f = 0; f < nf; ++f) {
                const real angle = base[k] * (real)(f + 1);
                out[k * nf * 2 + f * 2] = r_sin(angle); out[k * nf * 2 + f * 2 + 1] = r_cos(angle);
            }
    } else {
        for (int k = 0; k < ipd; ++k)
            for (int f = 0; f < nf; ++f) out[k * nf + f] = r_sin(base[k] * (real)ldexp(1.0, f));
    }
}
/* one hit of the feature forward (processHitParticle fwd with PerRayParticleFeatures, gutKBufferRenderer.cuh:199-225): the hit's weight, depth and
 * transmittance from (alpha, hitT) as tested - K > 0: as they sat in the hit buffer -, its features interpolated at the canonical
 * intersection, which is a function of (ray, particle) and is evaluated here with the operations of the test (density_hit_ex) */
static void nht_fwd_hit(const GutConfig* cfg, const int* nht, const orc_ray* ray, const real* density12, const real* features, const orc_hit* h,
                        int nr, real* T, real* D, real* acc, uint32_t* cnt, int* alive) {
    const real w = h->alpha * (*T);
    *D += h->hitT * w;
    *T *= (1 - h->alpha);
    if (w > 0) {
        const orc_particle p = load_particle(density12 + 12 * (size_t)h->idx);
        const v3 giscl = v3_make(1 / p.scl.x, 1 / p.scl.y, 1 / p.scl.z);
        const v3 gro = v3_mul(giscl, v3_mul_rows(v3_sub(ray->o, p.pos), &p.rotT));
        const v3 grdu = v3_mul(giscl, v3_mul_rows(ray->d, &p.rotT));
        const v3 grd = v3_scale(grdu, 1 / r_sqrt(v3_dot(grdu, grdu)));
        const v3 cg = v3_scale(grd, v3_dot(grd, v3_scale(gro, -1)));
        const v3 P = v3_add(gro, cg);
        real f[ORC_NHT_MAX_RAY_DIM];
        nht_features_at(nht, features + (size_t)nht[0] * h->idx, P, f);
        for (int i = 0; i < nr; ++i) acc[i] += f[i] * w;
        (*cnt)++;
    }
    if (*T < (real)cfg->min_transmittance) *alive = 0;
}
/* k_buffer_size > 0 (round 6): the sorted hit buffer of orc_gut_render_fwd (HitParticleKBufferT, :62-122, 331-351) in front of nht_fwd_hit */
int orc_gut_render_nht_fwd(const GutConfig* cfg, const int* nht, int width, int height, const real* pose_start7, const real* pose_end7,
                           const real* density12, const real* features, const uint32_t* sorted_idx, const uint32_t* tile_ranges,
                           const real* ray_o, const real* ray_d, real* out_fd, real* out_dist, real* out_cnt) {
    const orc_frame_poses fp = frame_poses(pose_start7, pose_end7);
    const int gx = tile_grid_dim(width);
    const int nr = nht_ray_dim(nht);
    const int K = cfg->k_buffer_size;
    if (K > ORC_MAX_K || nr > ORC_NHT_MAX_RAY_DIM || nht[1] > ORC_NHT_MAX_RAY_DIM) return -1;
#pragma omp parallel for schedule(dynamic, 16)
    for (int pix = 0; pix < width * height; ++pix) {
        const int x = pix % width, y = pix / width;
        const orc_ray ray = init_ray(&fp, ray_o + 3 * (size_t)pix, ray_d + 3 * (size_t)pix);
        if (!ray.valid) continue;
        const uint32_t tile = (uint32_t)((y / ORC_TILE) * gx + (x / ORC_TILE));
        const uint32_t beg = tile_ranges[2 * tile], end = tile_ranges[2 * tile + 1];
        real T = 1, D = 0, acc[ORC_NHT_MAX_RAY_DIM]; uint32_t cnt = 0; int alive = 1;
        for (int i = 0; i < nr; ++i) acc[i] = 0;
        orc_hit kbuf[ORC_MAX_K]; int nhits = 0;
        for (int k = 0; k < K; ++k) { kbuf[k].idx = ORC_INVALID_IDX; kbuf[k].hitT = -1; kbuf[k].alpha = 0; }
        for (uint32_t e = beg; e < end && alive; ++e) {
            const uint32_t idx = sorted_idx[e];
            if (idx == ORC_INVALID_IDX) break;
            const orc_particle p = load_particle(density12 + 12 * (size_t)idx);
            orc_hit h; h.idx = idx; h.hitT = -1; h.alpha = 0;
            if (!(density_hit(cfg, ray.o, ray.d, &p, &h.alpha, &h.hitT) && h.hitT > ray.tmin && h.hitT < ray.tmax)) continue;
            if (K == 0) {
                nht_fwd_hit(cfg, nht, &ray, density12, features, &h, nr, &T, &D, acc, &cnt, &alive);
            } else {
                if (nhits == K) {
                    nht_fwd_hit(cfg, nht, &ray, density12, features, &kbuf[0], nr, &T, &D, acc, &cnt, &alive);
                    kbuf[0].hitT = -1;
                } else nhits++;
                for (int i = K - 1; i >= 0; --i)
                    if (h.hitT > kbuf[i].hitT) { const orc_hit t = kbuf[i]; kbuf[i] = h; h = t; }
            }
        }
        if (K > 0)
            for (int i = 0; alive && i < nhits; ++i) nht_fwd_hit(cfg, nht, &ray, density12, features, &kbuf[K - nhits + i], nr, &T, &D, acc, &cnt, &alive);
        real* o = out_fd + (size_t)(nr + 1) * pix;
        for (int i = 0; i < nr; ++i) o[i] = acc[i];
        o[nr] = 1 - T;
        out_dist[pix] = D;
        if (cfg->enable_hitcounts) out_cnt[pix] = (real)cnt;
    }
    return 0;
}

/* orc_gut_pixel_trace for the feature path: additionally the nr feature values each entry's hit would blend (interpolated at ITS
 * canonical intersection - per ray, so they cannot come from a per-particle table like the SH radiance), out_feat [cap, nr]. */
int orc_gut_pixel_trace_nht(const GutConfig* cfg, const int* nht, int width, int height, const real* pose_start7, const real* pose_end7,
                            const real* density12, const real* features, const uint32_t* sorted_idx, const uint32_t* tile_ranges, const real* ray_o,
                            const real* ray_d, uint32_t pix, uint32_t cap, uint32_t* out_idx, real* out_alpha, real* out_hitT, real* out_margin,
                            real* out_feat) {
    const orc_frame_poses fp = frame_poses(pose_start7, pose_end7);
    const int gx = tile_grid_dim(width);
    const int nr = nht_ray_dim(nht);
    const int x = (int)pix % width, y = (int)pix / width;
    const orc_ray ray = init_ray(&fp, ray_o + 3 * (size_t)pix, ray_d + 3 * (size_t)pix);
    if (!ray.valid || nr > ORC_NHT_MAX_RAY_DIM) return 0;
    const uint32_t tile = (uint32_t)((y / ORC_TILE) * gx + (x / ORC_TILE));
    uint32_t n = 0;
    for (uint32_t e = tile_ranges[2 * tile]; e < tile_ranges[2 * tile + 1] && n < cap; ++e) {
        const uint32_t idx = sorted_idx[e];
        if (idx == ORC_INVALID_IDX) break;
        const orc_particle p = load_particle(density12 + 12 * (size_t)idx);
        /* as orc_gut_render_nht_fwd, every entry kept */
        const v3 giscl = v3_make(1 / p.scl.x, 1 / p.scl.y, 1 / p.scl.z);
        const v3 gro = v3_mul(giscl, v3_mul_rows(v3_sub(ray.o, p.pos), &p.rotT));
        const v3 grdu = v3_mul(giscl, v3_mul_rows(ray.d, &p.rotT));
        const v3 grd = v3_scale(grdu, 1 / r_sqrt(v3_dot(grdu, grdu)));
        const v3 gcrod = v3_cross(grd, gro);
        const real resp = particle_response(cfg->particle_kernel_degree, v3_dot(gcrod, gcrod));
        const real alpha = r_min((real)cfg->particle_kernel_max_alpha, resp * p.density);
        const v3 cg = v3_scale(grd, v3_dot(grd, v3_scale(gro, -1)));
        const v3 P = v3_add(gro, cg);
        const v3 grds = v3_mul(p.scl, cg);
        const real hitT = r_sqrt(v3_dot(grds, grds));
        real f[ORC_NHT_MAX_RAY_DIM];
        nht_features_at(nht, features + (size_t)nht[0] * idx, P, f);
        out_idx[n] = idx;
        out_alpha[n] = (hitT > ray.tmin && hitT < ray.tmax) ? alpha : 0;
        out_hitT[n] = hitT;
        out_margin[n] = r_min(resp / (real)cfg->particle_kernel_min_response, resp * p.density / (real)cfg->particle_kernel_min_alpha) - 1;
        for (int i = 0; i < nr; ++i) out_feat[(size_t)n * nr + i] = f[i];
        n++;
    }
    return (int)n;
}

static size_t list_particle_bound(int width, int height, const uint32_t* sorted_idx, const uint32_t* tile_ranges);
typedef struct {
    const GutConfig* cfg; const int* nht; const real* density12; const real* features; double* acc_d; double* acc_f;
    v3 v0, e1, e2, e3, c23, gw[4]; real inv_det; int nr, K, ipd, act, nf, points;
} nht_bwd_ctx;
typedef struct { real Cb[ORC_NHT_MAX_RAY_DIM], gC[ORC_NHT_MAX_RAY_DIM], Tb, gT, Db, gD, T; int alive; } nht_bwd_ray;
/* one hit of the feature backward (processHitParticle<Backward> with PerRayParticleFeatures, gutKBufferRenderer.cuh:158-198): the reverse of
 * nht_fwd_hit for the hit (particle, alpha, hitT) - K > 0: as it sat in the hit buffer; everything else is a function of (ray, particle) */
static void nht_bwd_hit(const nht_bwd_ctx* c, const orc_ray* rayp, nht_bwd_ray* st, const orc_hit* h) {
    const GutConfig* cfg = c->cfg; const int* nht = c->nht;
    const int nr = c->nr, K = c->K, ipd = c->ipd, act = c->act, nf = c->nf, points = c->points;
    const v3 v0 = c->v0, e1 = c->e1, e2 = c->e2, e3 = c->e3, c23 = c->c23; const v3* gw = c->gw; const real inv_det = c->inv_det;
    double* acc_d = c->acc_d; double* acc_f = c->acc_f;
    const orc_ray ray = *rayp;
    real* Cb = st->Cb; real* gC = st->gC;
    real Tb = st->Tb, gT = st->gT, Db = st->Db, gD = st->gD;
    const uint32_t idx = h->idx;
    const real alpha = h->alpha, hitT = h->hitT;
    const orc_particle p = load_particle(c->density12 + 12 * (size_t)idx);
    const real* features = c->features;
    const v3 gscl = p.scl;
    const v3 giscl = v3_make(1 / gscl.x, 1 / gscl.y, 1 / gscl.z);
    const v3 gposc = v3_sub(ray.o, p.pos);
    const v3 gposcr = v3_mul_rows(gposc, &p.rotT);
    const v3 gro = v3_mul(giscl, gposcr);
    const v3 rdr = v3_mul_rows(ray.d, &p.rotT);
    const v3 grdu = v3_mul(giscl, rdr);
    const v3 grd = v3_scale(grdu, 1 / r_sqrt(v3_dot(grdu, grdu)));
    const v3 gcrod = v3_cross(grd, gro);
    const real gray = v3_dot(gcrod, gcrod);
    const real gres = particle_response(cfg->particle_kernel_degree, gray);
    const real pdot = v3_dot(grd, v3_scale(gro, -1));
    const v3 grdd = v3_scale(grd, pdot);
    const v3 P = v3_add(gro, grdd);
    const v3 grds = v3_mul(gscl, grdd);
    const real gsq = v3_dot(grds, grds);
    /* ---- the hit's features and the reverse of their integration (lerp form, un-blending front to back) ---- */
    const real* row = features + (size_t)K * idx;
    real wq[4] = {1, 0, 0, 0};
    if (points == 4) {
        const v3 d = v3_sub(P, v0);
        wq[1] = v3_dot(d, c23) * inv_det; wq[2] = v3_dot(e1, v3_cross(d, e3)) * inv_det; wq[3] = v3_dot(e1, v3_cross(e2, d)) * inv_det;
        wq[0] = 1 - wq[1] - wq[2] - wq[3];
    }
    real base[ORC_NHT_MAX_RAY_DIM], f[ORC_NHT_MAX_RAY_DIM], gf[ORC_NHT_MAX_RAY_DIM], gbase[ORC_NHT_MAX_RAY_DIM];
    for (int n = 0; n < ipd; ++n) {
        base[n] = row[n] * wq[0];
        for (int k = 1; k < points; ++k) base[n] += wq[k] * row[k * ipd + n];
    }
    nht_features_at(nht, row, P, f);
    const real w = 1 / (1 - alpha);
    real dalpha = 0;
    if (alpha > 0) {   /* (particleFeaturesIntegrateBwdToBuffer: if (alpha > 0)) */
        for (int i = 0; i < nr; ++i) {
            Cb[i] = (Cb[i] - f[i] * alpha) * w;
            dalpha += (f[i] - Cb[i]) * gC[i];
            gf[i] = alpha * gC[i];
            gC[i] *= (1 - alpha);
        }
    } else {
        for (int i = 0; i < nr; ++i) gf[i] = 0;
    }
    /* activation backward */
    for (int n = 0; n < ipd; ++n) gbase[n] = 0;
    if (act == 0) { for (int i = 0; i < ipd; ++i) gbase[i] = gf[i]; }
    else if (act == 3) { for (int i = 0; i < ipd; ++i) gbase[i] = base[i] > 0 ? gf[i] : 0; }
    else if (act == 2) {
        for (int k = 0; k < ipd; ++k)
            for (int q = 0; q < nf; ++q) {
                const real fr = (real)(q + 1), ang = base[k] * fr;
                gbase[k] += fr * (r_cos(ang) * gf[k * nf * 2 + q * 2] - r_sin(ang) * gf[k * nf * 2 + q * 2 + 1]);
            }
    } else {
        for (int k = 0; k < ipd; ++k)
            for (int q = 0; q < nf; ++q) {
                const real fr = (real)ldexp(1.0, q);
                gbase[k] += fr * r_cos(base[k] * fr) * gf[k * nf + q];
            }
    }
    /* blend backward: feature rows and the canonical position */
    v3 dP = v3_make(0, 0, 0);
    for (int k = 0; k < points; ++k) {
        real dwk = 0;
        for (int n = 0; n < ipd; ++n) {
            const real g = wq[k] * gbase[n];
            if (g != 0) {
#pragma omp atomic
                acc_f[(size_t)K * idx + k * ipd + n] += (double)g;
            }
            dwk += row[k * ipd + n] * gbase[n];
        }
        if (points == 4) dP = v3_add(dP, v3_scale(gw[k], dwk));
    }
    /* ---- density: T_out = T_in (1 - alpha), D_front = lerp(D_behind, depth, alpha) (as process_hit_bwd_k) ---- */
    Tb *= w;
    Db = (Db - hitT * alpha) * w;
    dalpha += (hitT - Db) * gD - Tb * gT;
    const real ddepth = alpha * gD;
    gD *= (1 - alpha);
    gT *= (1 - alpha);
    real gd[12];
    for (int k = 0; k < 12; ++k) gd[k] = 0;
    real dres = 0, ddens = 0;
    if (gres * p.density < (real)cfg->particle_kernel_max_alpha) { dres = p.density * dalpha; ddens = gres * dalpha; }
    gd[3] = ddens;
    const real grayGrd = particle_response_grd(cfg->particle_kernel_degree, gray, gres, dres);
    const v3 grdsGrd = gsq > 0 ? v3_scale(grds, ddepth / hitT) : v3_make(0, 0, 0);
    const v3 gsclHit = v3_mul(grdd, grdsGrd);
    const real sdot = v3_dot(v3_mul(grdsGrd, gscl), grd);
    v3 grdHit = v3_sub(v3_scale(v3_mul(gscl, grdsGrd), pdot), v3_scale(gro, sdot));
    v3 groHit = v3_scale(grd, -sdot);
    /* canonical intersection P = gro + grd (grd . -gro) */
    const real gdP = v3_dot(grd, dP);
    groHit = v3_add(groHit, v3_sub(dP, v3_scale(grd, gdP)));
    grdHit = v3_add(grdHit, v3_sub(v3_scale(dP, pdot), v3_scale(gro, gdP)));
    const v3 gcrodGrd = v3_scale(gcrod, 2 * grayGrd);
    const v3 grdGrd = v3_make(gcrodGrd.z * gro.y - gcrodGrd.y * gro.z, gcrodGrd.x * gro.z - gcrodGrd.z * gro.x, gcrodGrd.y * gro.x - gcrodGrd.x * gro.y);
    const v3 groGrd = v3_make(gcrodGrd.y * grd.z - gcrodGrd.z * grd.y, gcrodGrd.z * grd.x - gcrodGrd.x * grd.z, gcrodGrd.x * grd.y - gcrodGrd.y * grd.x);
    const v3 groTot = v3_add(groGrd, groHit);
    const v3 gsclGro = v3_mul(v3_make(-gposcr.x / (gscl.x * gscl.x), -gposcr.y / (gscl.y * gscl.y), -gposcr.z / (gscl.z * gscl.z)), groTot);
    const v3 gposcrGrd = v3_mul(giscl, groTot);
    const v3 gposcGrd = matmul_bw_vec(&p.rotT, gposcrGrd);
    const v4 gq1 = matmul_bw_quat(gposc, gposcrGrd, p.quat);
    gd[0] = -gposcGrd.x; gd[1] = -gposcGrd.y; gd[2] = -gposcGrd.z;
    const v3 grduGrd = v3_safe_normalize_bw(grdu, v3_add(grdGrd, grdHit));
    const v3 sclGrd = v3_add(v3_add(gsclHit, gsclGro),
                             v3_mul(v3_make(-rdr.x / (gscl.x * gscl.x), -rdr.y / (gscl.y * gscl.y), -rdr.z / (gscl.z * gscl.z)), grduGrd));
    gd[8] = sclGrd.x; gd[9] = sclGrd.y; gd[10] = sclGrd.z;
    const v4 gq2 = matmul_bw_quat(ray.d, v3_mul(giscl, grduGrd), p.quat);
    gd[4] = gq1.x + gq2.x; gd[5] = gq1.y + gq2.y; gd[6] = gq1.z + gq2.z; gd[7] = gq1.w + gq2.w;
    for (int k = 0; k < 11; ++k)
        if (gd[k] != 0) {
#pragma omp atomic
            acc_d[12 * (size_t)idx + k] += (double)gd[k];
        }
    st->Tb = Tb; st->gT = gT; st->Db = Db; st->gD = gD;
    st->T *= (1 - alpha);
    if (st->T < (real)cfg->min_transmittance) st->alive = 0;
}
/* --------------------------------------------------------------------------------------
 * render backward with neural harmonic features (K = 0): evalBackwardNoKBuffer's PerRayParticleFeatures branch
 * (gutKBufferRenderer.cuh:546-641): per hit featuresIntegrateBwdToLocalGrad (Slang reverse mode of integrateFeaturesFromBuffer<true>,
 * neuralHarmonicFeaturesParticle.slang:198-228, 277-320: the ray state is un-blended front to back like in the K > 0 path) and
 * densityProcessHitBwdToBuffer with the gradient of the canonical intersection (gaussianParticles.slang:420-479).  Those are autodiff
 * products that are not in the checkout: like process_hit_bwd_k this is the reverse mode of the restated forward, checked against
 * float64 torch.autograd of that forward (tests/golden/autograd_gut_nht.npz, make_autograd_golden.py --nht).
 * fd [H,W,nr+1] = forward results, g_fd their upstream gradients; g_density12 [N,12] and g_features [N,K] are ADDED to. */
int orc_gut_render_nht_bwd(const GutConfig* cfg, const int* nht, int width, int height, const real* pose_start7, const real* pose_end7,
                           const real* density12, const real* features, const uint32_t* sorted_idx, const uint32_t* tile_ranges,
                           const real* ray_o, const real* ray_d, const real* fd, const real* g_fd, const real* dist, const real* g_dist,
                           real* g_density12, real* g_features) {
    const orc_frame_poses fp = frame_poses(pose_start7, pose_end7);
    const int gx = tile_grid_dim(width);
    const int nr = nht_ray_dim(nht), K = nht[0], ipd = nht[1], act = nht[3], nf = nht[4];
    const int points = nht[2] == 1 ? 4 : 1;
    const int Kb = cfg->k_buffer_size;   /* (K is the feature row length here) */
    if (Kb > ORC_MAX_K || nr > ORC_NHT_MAX_RAY_DIM || ipd > ORC_NHT_MAX_RAY_DIM) return -1;
    const size_t n_acc = list_particle_bound(width, height, sorted_idx, tile_ranges);
    double* acc_d = (double*)calloc(n_acc * 12 + 1, sizeof(double));
    double* acc_f = (double*)calloc(n_acc * (size_t)K + 1, sizeof(double));
    if (!acc_d || !acc_f) { free(acc_d); free(acc_f); return -2; }
    /* canonical tetrahedron: gradients of the barycentric weights (constant vectors) */
    const real edge = R_(4.898979485566356), face_h = R_(4.242640687119285), face_in = R_(1.4142135623730951);
    const v3 v0 = v3_make(R_(0.5) * edge, -face_in, R_(-1.0)), v1 = v3_make(R_(-0.5) * edge, -face_in, R_(-1.0));
    const v3 v2 = v3_make(0, face_h - face_in, R_(-1.0)), v3_ = v3_make(0, 0, R_(3.0));
    const v3 e1 = v3_sub(v1, v0), e2 = v3_sub(v2, v0), e3 = v3_sub(v3_, v0);
    const v3 c23 = v3_cross(e2, e3);
    const real inv_det = 1 / v3_dot(e1, c23);
    v3 gw[4];
    gw[1] = v3_scale(c23, inv_det); gw[2] = v3_scale(v3_cross(e3, e1), inv_det); gw[3] = v3_scale(v3_cross(e1, e2), inv_det);
    gw[0] = v3_scale(v3_add(v3_add(gw[1], gw[2]), gw[3]), -1);
    nht_bwd_ctx ctx;
    ctx.cfg = cfg; ctx.nht = nht; ctx.density12 = density12; ctx.features = features; ctx.acc_d = acc_d; ctx.acc_f = acc_f;
    ctx.v0 = v0; ctx.e1 = e1; ctx.e2 = e2; ctx.e3 = e3; ctx.c23 = c23; for (int k = 0; k < 4; ++k) ctx.gw[k] = gw[k];
    ctx.inv_det = inv_det; ctx.nr = nr; ctx.K = K; ctx.ipd = ipd; ctx.act = act; ctx.nf = nf; ctx.points = points;
#pragma omp parallel for schedule(dynamic, 16)
    for (int pix = 0; pix < width * height; ++pix) {
        const int x = pix % width, y = pix / width;
        const orc_ray ray = init_ray(&fp, ray_o + 3 * (size_t)pix, ray_d + 3 * (size_t)pix);
        if (!ray.valid) continue;
        const uint32_t tile = (uint32_t)((y / ORC_TILE) * gx + (x / ORC_TILE));
        const uint32_t beg = tile_ranges[2 * tile], end = tile_ranges[2 * tile + 1];
        nht_bwd_ray st;
        const real* f_in = fd + (size_t)(nr + 1) * pix;
        const real* g_in = g_fd + (size_t)(nr + 1) * pix;
        for (int i = 0; i < nr; ++i) { st.Cb[i] = f_in[i]; st.gC[i] = g_in[i]; }
        st.Tb = 1 - f_in[nr]; st.gT = -g_in[nr]; st.Db = dist[pix]; st.gD = g_dist ? g_dist[pix] : 0;
        st.T = 1; st.alive = 1;
        orc_hit kbuf[ORC_MAX_K]; int nhits = 0;
        for (int k = 0; k < Kb; ++k) { kbuf[k].idx = ORC_INVALID_IDX; kbuf[k].hitT = -1; kbuf[k].alpha = 0; }
        for (uint32_t e = beg; e < end && st.alive; ++e) {
            const uint32_t idx = sorted_idx[e];
            if (idx == ORC_INVALID_IDX) break;
            const orc_particle p = load_particle(density12 + 12 * (size_t)idx);
            orc_hit h; h.idx = idx; h.hitT = -1; h.alpha = 0;
            if (!(density_hit(cfg, ray.o, ray.d, &p, &h.alpha, &h.hitT) && h.hitT > ray.tmin && h.hitT < ray.tmax)) continue;
            if (Kb == 0) {
                nht_bwd_hit(&ctx, &ray, &st, &h);
            } else {   /* the sorted hit buffer of render_bwd_kbuffer */
                if (nhits == Kb) {
                    nht_bwd_hit(&ctx, &ray, &st, &kbuf[0]);
                    kbuf[0].hitT = -1;
                } else nhits++;
                for (int i = Kb - 1; i >= 0; --i)
                    if (h.hitT > kbuf[i].hitT) { const orc_hit t = kbuf[i]; kbuf[i] = h; h = t; }
            }
        }
        if (Kb > 0)
            for (int i = 0; st.alive && i < nhits; ++i) nht_bwd_hit(&ctx, &ray, &st, &kbuf[Kb - nhits + i]);
    }
    for (size_t k = 0; k < n_acc * 12; ++k) g_density12[k] += (real)acc_d[k];
    for (size_t k = 0; k < n_acc * (size_t)K; ++k) g_features[k] += (real)acc_f[k];
    free(acc_d); free(acc_f);
    return 0;
}

/* Analysis aid (scripts/slab_analysis.py): per pixel, how many entries of its tile list the K = 0 forward loop examines
 * before the ray ends (the whole list if it never does).  Same loop as orc_gut_render_fwd. */
int orc_gut_render_fwd_consumed(const GutConfig* cfg, int width, int height, const real* pose_start7, const real* pose_end7,
                                const real* density12, const uint32_t* sorted_idx, const uint32_t* tile_ranges, const real* ray_o,
                                const real* ray_d, uint32_t* out_consumed) {
    const orc_frame_poses fp = frame_poses(pose_start7, pose_end7);
    const int gx = tile_grid_dim(width);
#pragma omp parallel for schedule(dynamic, 16)
    for (int pix = 0; pix < width * height; ++pix) {
        const int x = pix % width, y = pix / width;
        const orc_ray ray = init_ray(&fp, ray_o + 3 * (size_t)pix, ray_d + 3 * (size_t)pix);
        out_consumed[pix] = 0;
        if (!ray.valid) continue;
        const uint32_t tile = (uint32_t)((y / ORC_TILE) * gx + (x / ORC_TILE));
        const uint32_t beg = tile_ranges[2 * tile], end = tile_ranges[2 * tile + 1];
        real T = 1;
        uint32_t e = beg;
        for (; e < end; ++e) {
            const uint32_t idx = sorted_idx[e];
            if (idx == ORC_INVALID_IDX) break;
            const orc_particle p = load_particle(density12 + 12 * (size_t)idx);
            real alpha, hitT;
            if (density_hit(cfg, ray.o, ray.d, &p, &alpha, &hitT) && hitT > ray.tmin && hitT < ray.tmax) {
                T *= (1 - alpha);
                if (T < (real)cfg->min_transmittance) { ++e; break; }
            }
        }
        out_consumed[pix] = e - beg;
    }
    return 0;
}

/* Threshold analysis for the full-size parity tests (tests/parity_util.py).  For each listed pixel the K = 0 forward loop is
 * replayed and the evaluated entries that lie within a relative margin of one of the algorithm's discontinuities are counted:
 * response vs min_response, alpha vs min_alpha, the running transmittance vs min_transmittance.  A pixel with fewer than two
 * such entries cannot have had two opposite accept / reject flips, i.e. cannot differ with an unchanged hit count. */
int orc_gut_pixel_margins(const GutConfig* cfg, int width, int height, const real* pose_start7, const real* pose_end7,
                          const real* density12, const uint32_t* sorted_idx, const uint32_t* tile_ranges, const real* ray_o,
                          const real* ray_d, uint32_t npix, const uint32_t* pix_ids, real rel_margin, uint32_t* out_borderline) {
    const orc_frame_poses fp = frame_poses(pose_start7, pose_end7);
    const int gx = tile_grid_dim(width);
    const real min_resp = (real)cfg->particle_kernel_min_response, min_alpha = (real)cfg->particle_kernel_min_alpha;
    const real min_T = (real)cfg->min_transmittance;
#pragma omp parallel for schedule(dynamic, 4)
    for (uint32_t k = 0; k < npix; ++k) {
        const int pix = (int)pix_ids[k];
        const int x = pix % width, y = pix / width;
        out_borderline[k] = 0;
        const orc_ray ray = init_ray(&fp, ray_o + 3 * (size_t)pix, ray_d + 3 * (size_t)pix);
        if (!ray.valid) continue;
        const uint32_t tile = (uint32_t)((y / ORC_TILE) * gx + (x / ORC_TILE));
        real T = 1;
        uint32_t n = 0;
        for (uint32_t e = tile_ranges[2 * tile]; e < tile_ranges[2 * tile + 1]; ++e) {
            const uint32_t idx = sorted_idx[e];
            if (idx == ORC_INVALID_IDX) break;
            const orc_particle p = load_particle(density12 + 12 * (size_t)idx);
            real alpha, hitT, resp;
            const int acc = density_hit_ex(cfg, ray.o, ray.d, &p, &alpha, &hitT, &resp);
            const real a_raw = resp * p.density;
            /* the accept test is (resp > min_resp) && (alpha > min_alpha): the binding one of the two decides */
            const real m1 = r_fabs(resp / min_resp - 1), m2 = r_fabs(a_raw / min_alpha - 1);
            const int near_accept = (resp > min_resp * (1 - rel_margin) && a_raw > min_alpha * (1 - rel_margin)) &&
                                    (m1 < rel_margin || m2 < rel_margin);
            if (near_accept) n++;
            if (acc && hitT > ray.tmin && hitT < ray.tmax) {
                T *= (1 - alpha);
                if (r_fabs(T / min_T - 1) < 100 * rel_margin) n++;   /* T carries the rounding of every factor before it */
                if (T < min_T) break;
            }
        }
        out_borderline[k] = n;
    }
    return 0;
}

/* Per-entry trace of ONE pixel's tile list for the flip identification of the full-size parity tests: for every entry of the
 * tile (no early termination: a toggled decision may move the end) the particle, the alpha the hit would composite with
 * (min(max_alpha, resp * density), 0 when the hit distance is outside the ray interval), the hit distance, and the signed
 * relative margin of the accept test ( min(resp / min_response, resp * density / min_alpha) - 1 : accepted iff > 0 ).
 * Returns the number of entries written (<= cap). */
int orc_gut_pixel_trace(const GutConfig* cfg, int width, int height, const real* pose_start7, const real* pose_end7,
                        const real* density12, const uint32_t* sorted_idx, const uint32_t* tile_ranges, const real* ray_o,
                        const real* ray_d, uint32_t pix, uint32_t cap, uint32_t* out_idx, real* out_alpha, real* out_hitT, real* out_margin) {
    const orc_frame_poses fp = frame_poses(pose_start7, pose_end7);
    const int gx = tile_grid_dim(width);
    const int x = (int)pix % width, y = (int)pix / width;
    const orc_ray ray = init_ray(&fp, ray_o + 3 * (size_t)pix, ray_d + 3 * (size_t)pix);
    if (!ray.valid) return 0;
    const uint32_t tile = (uint32_t)((y / ORC_TILE) * gx + (x / ORC_TILE));
    uint32_t n = 0;
    for (uint32_t e = tile_ranges[2 * tile]; e < tile_ranges[2 * tile + 1] && n < cap; ++e) {
        const uint32_t idx = sorted_idx[e];
        if (idx == ORC_INVALID_IDX) break;
        const orc_particle p = load_particle(density12 + 12 * (size_t)idx);
        real alpha, hitT = 0, resp;
        density_hit_ex(cfg, ray.o, ray.d, &p, &alpha, &hitT, &resp);
        {   /* the hit distance of a rejected entry too (density_hit_ex only fills it on accept) */
            GutConfig all = *cfg;
            all.particle_kernel_min_response = 0; all.particle_kernel_min_alpha = -1;
            real a2, r2;
            density_hit_ex(&all, ray.o, ray.d, &p, &a2, &hitT, &r2);
        }
        const real m = r_min(resp / (real)cfg->particle_kernel_min_response, resp * p.density / (real)cfg->particle_kernel_min_alpha) - 1;
        out_idx[n] = idx;
        out_alpha[n] = (hitT > ray.tmin && hitT < ray.tmax) ? alpha : 0;
        out_hitT[n] = hitT;
        out_margin[n] = m;
        n++;
    }
    return (int)n;
}

/* --------------------------------------------------------------------------------------
 * render backward (K=0, SH branch) — gutKBufferRenderer.cuh:642-716, rayPayloadBackward.cuh:30-73
 * g_density12 [N,12] and g_rgb [N,3] are accumulated into (must arrive zeroed).
 * ------------------------------------------------------------------------------------ */
/* 1 + the largest particle index that occurs in the tile lists (the backward entry points are not told N) */
static size_t list_particle_bound(int width, int height, const uint32_t* sorted_idx, const uint32_t* tile_ranges) {
    const int tiles = tile_grid_dim(width) * tile_grid_dim(height);
    size_t bound = 0;
    for (int t = 0; t < tiles; ++t)
        for (uint32_t e = tile_ranges[2 * t]; e < tile_ranges[2 * t + 1]; ++e)
            if (sorted_idx[e] != ORC_INVALID_IDX && (size_t)sorted_idx[e] + 1 > bound) bound = (size_t)sorted_idx[e] + 1;
    return bound;
}

/* evalKBuffer<Backward> (gutKBufferRenderer.cuh:273-352): the forward's k-buffer walk, each popped hit differentiated */
static int render_bwd_kbuffer(const GutConfig* cfg, int width, int height, const real* pose_start7, const real* pose_end7,
                              const real* density12, const real* rgb, const uint32_t* sorted_idx, const uint32_t* tile_ranges,
                              const real* ray_o, const real* ray_d, const real* fd, const real* g_fd, const real* dist, const real* g_dist,
                              real* g_density12, real* g_rgb) {
    const orc_frame_poses fp = frame_poses(pose_start7, pose_end7);
    const int gx = tile_grid_dim(width);
    const int K = cfg->k_buffer_size;
    if (K > ORC_MAX_K) return -1;
    const size_t n_acc = list_particle_bound(width, height, sorted_idx, tile_ranges);   /* sums in double, see orc_gut_render_bwd */
    double* acc_d = (double*)calloc(n_acc * 12 + 1, sizeof(double));
    double* acc_c = (double*)calloc(n_acc * 3 + 1, sizeof(double));
#pragma omp parallel for schedule(dynamic, 16)
    for (int pix = 0; pix < width * height; ++pix) {
        const int x = pix % width, y = pix / width;
        const orc_ray ray = init_ray(&fp, ray_o + 3 * (size_t)pix, ray_d + 3 * (size_t)pix);
        if (!ray.valid) continue;
        const uint32_t tile = (uint32_t)((y / ORC_TILE) * gx + (x / ORC_TILE));
        const uint32_t beg = tile_ranges[2 * tile], end = tile_ranges[2 * tile + 1];
        orc_kbwd_ray r;
        r.Cb = v3_make(fd[4 * (size_t)pix], fd[4 * (size_t)pix + 1], fd[4 * (size_t)pix + 2]);
        r.gC = v3_make(g_fd[4 * (size_t)pix], g_fd[4 * (size_t)pix + 1], g_fd[4 * (size_t)pix + 2]);
        r.Tb = 1 - fd[4 * (size_t)pix + 3];
        r.gT = -g_fd[4 * (size_t)pix + 3];
        r.Db = dist[pix]; r.gD = g_dist[pix];
        real T = 1; int alive = 1;
        orc_hit kbuf[ORC_MAX_K]; int nhits = 0;
        for (int k = 0; k < K; ++k) { kbuf[k].idx = ORC_INVALID_IDX; kbuf[k].hitT = -1; kbuf[k].alpha = 0; }
#define ORC_KBWD_PROCESS(h)                                                                                             \
        do {                                                                                                                \
            const orc_particle pp = load_particle(density12 + 12 * (size_t)(h).idx);                                        \
            const real* c = rgb + 3 * (size_t)(h).idx;                                                                      \
            real gd[12], gf[3];                                                                                             \
            process_hit_bwd_k(cfg, ray.o, ray.d, &pp, v3_make(r_max(c[0], 0), r_max(c[1], 0), r_max(c[2], 0)), (h).alpha, (h).hitT, &r, gd, gf); \
            for (int k = 0; k < 11; ++k) if (gd[k] != 0) { _Pragma("omp atomic") acc_d[12 * (size_t)(h).idx + k] += (double)gd[k]; }           \
            for (int k = 0; k < 3; ++k) if (gf[k] != 0) { _Pragma("omp atomic") acc_c[3 * (size_t)(h).idx + k] += (double)gf[k]; }             \
            T *= (1 - (h).alpha);                                                                                           \
            if (T < (real)cfg->min_transmittance) alive = 0;                                                                \
        } while (0)
        for (uint32_t e = beg; e < end && alive; ++e) {
            const uint32_t idx = sorted_idx[e];
            if (idx == ORC_INVALID_IDX) break;
            const orc_particle p = load_particle(density12 + 12 * (size_t)idx);
            orc_hit h; h.idx = idx; h.hitT = -1; h.alpha = 0;
            if (density_hit(cfg, ray.o, ray.d, &p, &h.alpha, &h.hitT) && h.hitT > ray.tmin && h.hitT < ray.tmax) {
                const int full = nhits == K;
                if (full) {
                    ORC_KBWD_PROCESS(kbuf[0]);
                    kbuf[0].hitT = -1;
                } else nhits++;
                for (int i = K - 1; i >= 0; --i)
                    if (h.hitT > kbuf[i].hitT) { const orc_hit t = kbuf[i]; kbuf[i] = h; h = t; }
            }
        }
        for (int i = 0; alive && i < nhits; ++i) ORC_KBWD_PROCESS(kbuf[K - nhits + i]);
#undef ORC_KBWD_PROCESS
    }
    for (size_t k = 0; k < n_acc * 12; ++k) g_density12[k] += (real)acc_d[k];
    for (size_t k = 0; k < n_acc * 3; ++k) g_rgb[k] += (real)acc_c[k];
    free(acc_d); free(acc_c);
    return 0;
}

int orc_gut_render_bwd(const GutConfig* cfg, int width, int height, const real* pose_start7, const real* pose_end7,
                       const real* density12, const real* rgb, const uint32_t* sorted_idx, const uint32_t* tile_ranges,
                       const real* ray_o, const real* ray_d,
                       const real* fd, const real* g_fd, const real* dist, const real* g_dist,
                       real* g_density12, real* g_rgb) {
    if (cfg->k_buffer_size != 0)
        return render_bwd_kbuffer(cfg, width, height, pose_start7, pose_end7, density12, rgb, sorted_idx, tile_ranges, ray_o, ray_d, fd, g_fd, dist,
                                  g_dist, g_density12, g_rgb);
    const orc_frame_poses fp = frame_poses(pose_start7, pose_end7);
    const int gx = tile_grid_dim(width);
    /* Per-hit gradients are computed in `real` (the reference's arithmetic type) but SUMMED in double: the reference adds them
     * with float atomics in an order that changes from run to run, so its own sums carry an order-dependent rounding noise; the
     * checker removes that noise instead of reproducing one sample of it. */
    const size_t n_acc = list_particle_bound(width, height, sorted_idx, tile_ranges);
    double* acc_d = (double*)calloc(n_acc * 12 + 1, sizeof(double));
    double* acc_c = (double*)calloc(n_acc * 3 + 1, sizeof(double));
#pragma omp parallel for schedule(dynamic, 16)
    for (int pix = 0; pix < width * height; ++pix) {
        const int x = pix % width, y = pix / width;
        const orc_ray ray = init_ray(&fp, ray_o + 3 * (size_t)pix, ray_d + 3 * (size_t)pix);
        if (!ray.valid) continue;
        const uint32_t tile = (uint32_t)((y / ORC_TILE) * gx + (x / ORC_TILE));
        const uint32_t beg = tile_ranges[2 * tile], end = tile_ranges[2 * tile + 1];
        orc_bwd_ray r;
        r.T = 1; r.feat = v3_make(0, 0, 0); r.hitT = 0;
        r.feat_fin = v3_make(fd[4 * (size_t)pix], fd[4 * (size_t)pix + 1], fd[4 * (size_t)pix + 2]);
        r.feat_grad = v3_make(g_fd[4 * (size_t)pix], g_fd[4 * (size_t)pix + 1], g_fd[4 * (size_t)pix + 2]);
        r.T_fin = 1 - fd[4 * (size_t)pix + 3];
        r.T_grad = -g_fd[4 * (size_t)pix + 3];
        r.hitT_fin = dist[pix]; r.hitT_grad = g_dist[pix];
        for (uint32_t e = beg; e < end; ++e) {
            const uint32_t idx = sorted_idx[e];
            if (idx == ORC_INVALID_IDX) break;
            const orc_particle p = load_particle(density12 + 12 * (size_t)idx);
            const real* c = rgb + 3 * (size_t)idx;
            real gd[12] = {0}, gf[3] = {0};
            process_hit_bwd(cfg, ray.o, ray.d, &p, v3_make(r_max(c[0], 0), r_max(c[1], 0), r_max(c[2], 0)), &r, gd, gf);
            for (int k = 0; k < 11; ++k) {
                if (gd[k] != 0) {
#pragma omp atomic
                    acc_d[12 * (size_t)idx + k] += (double)gd[k];
                }
            }
            for (int k = 0; k < 3; ++k) {
                if (gf[k] != 0) {
#pragma omp atomic
                    acc_c[3 * (size_t)idx + k] += (double)gf[k];
                }
            }
            if (r.T < (real)cfg->min_transmittance) break;
        }
    }
    for (size_t k = 0; k < n_acc * 12; ++k) g_density12[k] += (real)acc_d[k];
    for (size_t k = 0; k < n_acc * 3; ++k) g_rgb[k] += (real)acc_c[k];
    free(acc_d); free(acc_c);
    return 0;
}

/* --------------------------------------------------------------------------------------
 * projection backward — gutProjector.cuh:390-430; sphericalHarmonics.slang:21-64 (bwd),
 * gaussianParticles.slang:320-340,545-558 (incident direction bwd)
 * g_sph [N,ncoef*3] is overwritten for every particle; g_density12 position is accumulated.
 * ------------------------------------------------------------------------------------ */
int orc_gut_project_bwd(const GutConfig* cfg, const real* pose_start7, const real* pose_end7, uint32_t N, int n_active_features,
                        const uint32_t* tiles_count, const real* density12, const real* sph, const real* g_rgb,
                        real* g_density12, real* g_sph) {
    const orc_frame_poses fp = frame_poses(pose_start7, pose_end7);
    const int ncoef = (cfg->particle_radiance_sph_degree + 1) * (cfg->particle_radiance_sph_degree + 1);
    const int nact = (n_active_features + 1) * (n_active_features + 1);
    for (uint32_t i = 0; i < N; ++i) {
        real* gs = g_sph + (size_t)i * 3 * ncoef;
        for (int k = 0; k < 3 * ncoef; ++k) gs[k] = 0;
        if (tiles_count[i] == 0) continue;
        const real* pd = density12 + 12 * (size_t)i;
        const v3 v = v3_sub(v3_make(pd[0], pd[1], pd[2]), fp.sensor_world_pos);
        const real len = r_sqrt(v3_dot(v, v));
        const v3 dir = v3_scale(v, 1 / len);
        const real* coeffs = sph + (size_t)i * 3 * ncoef;
        const v3 cu = sh_radiance_unclamped(n_active_features, coeffs, dir);
        v3 g = v3_make(g_rgb[3 * i], g_rgb[3 * i + 1], g_rgb[3 * i + 2]);
        if (!(cu.x > 0)) g.x = 0;
        if (!(cu.y > 0)) g.y = 0;
        if (!(cu.z > 0)) g.z = 0;
        real b[16]; v3 db[16];
        sh_basis(n_active_features, dir, b);
        sh_basis_grad(n_active_features, dir, db);
        v3 gdir = v3_make(0, 0, 0);
        for (int k = 0; k < nact && k < ncoef; ++k) {
            gs[3 * k] = b[k] * g.x; gs[3 * k + 1] = b[k] * g.y; gs[3 * k + 2] = b[k] * g.z;
            const real s = g.x * coeffs[3 * k] + g.y * coeffs[3 * k + 1] + g.z * coeffs[3 * k + 2];
            gdir = v3_add(gdir, v3_scale(db[k], s));
        }
        /* normalize backward: (g - n (n.g)) / |v| */
        const real ng = v3_dot(dir, gdir);
        const v3 gpos = v3_scale(v3_sub(gdir, v3_scale(dir, ng)), 1 / len);
        g_density12[12 * (size_t)i] += gpos.x; g_density12[12 * (size_t)i + 1] += gpos.y; g_density12[12 * (size_t)i + 2] += gpos.z;
    }
    return 0;
}

/* known-answer entry points for the camera / pose functions (pinned by tests/golden/camera.npz, which the reference's own
 * cameraProjections.cuh + sensors.h produced on the host: oracle/ref/ref_camera.cpp) */
int orc_kat_project_point_with_shutter(const GrutCamera* cam, const real* pose_start7, const real* pose_end7, int n_iter, const real* pos3,
                                       real tol, real* out2) {
    return project_point_with_shutter(cam, pose_from7(pose_start7), pose_from7(pose_end7), n_iter, v3_make(pos3[0], pos3[1], pos3[2]), tol,
                                      &out2[0], &out2[1]);
}
static void pose_to7(orc_pose p, real* o) { o[0] = p.t.x; o[1] = p.t.y; o[2] = p.t.z; o[3] = p.q.x; o[4] = p.q.y; o[5] = p.q.z; o[6] = p.q.w; }
void orc_kat_pose_inverse(const real* p7, real* out7) { pose_to7(pose_inverse(pose_from7(p7)), out7); }
void orc_kat_pose_interpolate(const real* a7, const real* b7, real t, real* out7) {
    pose_to7(pose_interpolate(pose_from7(a7), pose_from7(b7), t), out7);
}

int orc_sizeof_real(void) { return (int)sizeof(real); }
