/*
 * orc_math.h — shared math of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * The oracle is a CPU restatement of the reference's algorithms, used solely by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker
 * for the HIP path.  Nothing in the product package may include or load it.
 *
 * Compiled twice: -DORC_REAL=float (liboracle32.so) and -DORC_REAL=double
 * (liboracle64.so, used for finite-difference gradient checks).
 *
 * Each function cites the reference file:line (under /root/reference) it restates.
 * Third-party arithmetic that is absent from /root/reference (tiny-cuda-nn
 * vec/mat/quat helpers: to_mat3, quat(mat3), slerp, mix; un-vendored submodule
 * thirdparty/tiny-cuda-nn, no pinned SHA available) is restated here from its
 * published definition (glm-compatible, column-major mat3, quat ctor (w,x,y,z)).
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef ORC_REAL
#define ORC_REAL float
#endif
typedef ORC_REAL real;

#define R_(x) ((real)(x))

static inline real r_exp(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }
static inline real r_log(real x) { return sizeof(real) == 4 ? (real)logf((float)x) : (real)log((double)x); }
static inline real r_sqrt(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
static inline real r_pow(real x, real y) { return sizeof(real) == 4 ? (real)powf((float)x, (float)y) : (real)pow((double)x, (double)y); }
static inline real r_atan2(real y, real x) { return sizeof(real) == 4 ? (real)atan2f((float)y, (float)x) : (real)atan2((double)y, (double)x); }
static inline real r_floor(real x) { return sizeof(real) == 4 ? (real)floorf((float)x) : (real)floor((double)x); }
static inline real r_ceil(real x) { return sizeof(real) == 4 ? (real)ceilf((float)x) : (real)ceil((double)x); }
static inline real r_fabs(real x) { return x < 0 ? -x : x; }
/* fused multiply-add and IEEE minNum / maxNum in the oracle's arithmetic type (what v_fma_f32 / v_min_f32 / v_max_f32 compute) */
/* (the hardware instruction where the host has it — same correctly rounded result as libm's software fmaf, 20x faster) */
#if defined(__FMA__)   /* built with -mfma (oracle/Makefile does when the build host has the unit): the instruction, inline */
static inline float orc_fmaf_hw(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline double orc_fma_hw(double a, double b, double c) { return __builtin_fma(a, b, c); }
static inline int orc_have_fma(void) { return 1; }
#elif defined(__x86_64__) && defined(__GNUC__)
__attribute__((target("fma"))) static float orc_fmaf_hw(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__attribute__((target("fma"))) static double orc_fma_hw(double a, double b, double c) { return __builtin_fma(a, b, c); }
static inline int orc_have_fma(void) { static int have = -1; if (have < 0) have = __builtin_cpu_supports("fma") ? 1 : 0; return have; }
#else
static float orc_fmaf_hw(float a, float b, float c) { return fmaf(a, b, c); }
static double orc_fma_hw(double a, double b, double c) { return fma(a, b, c); }
static inline int orc_have_fma(void) { return 0; }
#endif
static inline real r_fma(real a, real b, real c) {
    if (sizeof(real) == 4) return (real)(orc_have_fma() ? orc_fmaf_hw((float)a, (float)b, (float)c) : fmaf((float)a, (float)b, (float)c));
    return (real)(orc_have_fma() ? orc_fma_hw((double)a, (double)b, (double)c) : fma((double)a, (double)b, (double)c));
}
static inline real r_fmin(real a, real b) { return sizeof(real) == 4 ? (real)fminf((float)a, (float)b) : (real)fmin((double)a, (double)b); }
static inline real r_fmax(real a, real b) { return sizeof(real) == 4 ? (real)fmaxf((float)a, (float)b) : (real)fmax((double)a, (double)b); }
static inline real r_min(real a, real b) { return a < b ? a : b; }
static inline real r_max(real a, real b) { return a > b ? a : b; }
static inline real r_sin(real x) { return (real)sin((double)x); }
static inline real r_cos(real x) { return (real)cos((double)x); }
static inline real r_acos(real x) { return (real)acos((double)x); }
static inline real r_hypot(real x, real y) { return (real)hypot((double)x, (double)y); }

typedef struct { real x, y, z; } v3;
typedef struct { real x, y, z, w; } v4;
typedef struct { v3 r[3]; } m33; /* three rows */

static inline v3 v3_make(real x, real y, real z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, real s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline real v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* mathUtils.cuh:380-383 */
static inline v3 v3_safe_normalize(v3 v) {
    const real l = v3_dot(v, v);
    return l > 0 ? v3_scale(v, R_(1) / r_sqrt(l)) : v;
}
/* mathUtils.cuh:413-424 safe_normalize_bw */
static inline v3 v3_safe_normalize_bw(v3 v, v3 d_out) {
    const real l = v3_dot(v, v);
    if (l > 0) {
        const real il  = R_(1) / r_sqrt(l);
        const real il3 = il * il * il;
        const real s   = d_out.x * v.x + d_out.y * v.y + d_out.z * v.z;
        return v3_make(il * d_out.x - il3 * s * v.x, il * d_out.y - il3 * s * v.y, il * d_out.z - il3 * s * v.z);
    }
    return v3_make(0, 0, 0);
}

/* models/gaussianParticles.cuh:39-59 quaternionWXYZToMatrix: ret[i] are the ROWS of R^T
 * (= columns of the standard rotation R).  p * ret == (ret[0].p, ret[1].p, ret[2].p) == R^T p. */
static inline m33 quat_wxyz_to_rotT(v4 q /* x=w(r), y=x, z=y, w=z as stored [w,x,y,z] */) {
    const real r = q.x, x = q.y, y = q.z, z = q.w;
    const real xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
    const real rx = r * x, ry = r * y, rz = r * z;
    m33 m;
    m.r[0] = v3_make(R_(1) - R_(2) * (yy + zz), R_(2) * (xy + rz), R_(2) * (xz - ry));
    m.r[1] = v3_make(R_(2) * (xy - rz), R_(1) - R_(2) * (xx + zz), R_(2) * (yz + rx));
    m.r[2] = v3_make(R_(2) * (xz + ry), R_(2) * (yz - rx), R_(1) - R_(2) * (xx + yy));
    return m;
}
/* p * M  (mathUtils.cuh:447-449): rows dot p */
static inline v3 v3_mul_rows(v3 p, const m33* m) {
    return v3_make(v3_dot(m->r[0], p), v3_dot(m->r[1], p), v3_dot(m->r[2], p));
}
/* M * p  (mathUtils.cuh:440-445): columns dot p */
static inline v3 m33_mul_cols(const m33* m, v3 p) {
    return v3_make(m->r[0].x * p.x + m->r[1].x * p.y + m->r[2].x * p.z,
                   m->r[0].y * p.x + m->r[1].y * p.y + m->r[2].y * p.z,
                   m->r[0].z * p.x + m->r[1].z * p.y + m->r[2].z * p.z);
}
/* mathUtils.cuh:451-456 matmul_bw_vec */
static inline v3 matmul_bw_vec(const m33* m, v3 g) {
    return v3_make(g.x * m->r[0].x + g.y * m->r[1].x + g.z * m->r[2].x,
                   g.x * m->r[0].y + g.y * m->r[1].y + g.z * m->r[2].y,
                   g.x * m->r[0].z + g.y * m->r[1].z + g.z * m->r[2].z);
}
/* mathUtils.cuh:458-521 matmul_bw_quat: gradient of (p * rotT(q)) w.r.t. q=(r,x,y,z), upstream g */
static inline v4 matmul_bw_quat(v3 p, v3 g, v4 q) {
    const v3 d0 = v3_scale(p, g.x), d1 = v3_scale(p, g.y), d2 = v3_scale(p, g.z);
    const real r = q.x, x = q.y, y = q.z, z = q.w;
    real dr = 0, dx = 0, dy = 0, dz = 0;
    dy += -4 * y * d0.x; dz += -4 * z * d0.x;
    dr += 2 * z * d0.y; dx += 2 * y * d0.y; dy += 2 * x * d0.y; dz += 2 * r * d0.y;
    dr += -2 * y * d0.z; dx += 2 * z * d0.z; dy += -2 * r * d0.z; dz += 2 * x * d0.z;
    dr += -2 * z * d1.x; dx += 2 * y * d1.x; dy += 2 * x * d1.x; dz += -2 * r * d1.x;
    dx += -4 * x * d1.y; dz += -4 * z * d1.y;
    dr += 2 * x * d1.z; dx += 2 * r * d1.z; dy += 2 * z * d1.z; dz += 2 * y * d1.z;
    dr += 2 * y * d2.x; dx += 2 * z * d2.x; dy += 2 * r * d2.x; dz += 2 * x * d2.x;
    dr += -2 * x * d2.y; dx += -2 * r * d2.y; dy += 2 * z * d2.y; dz += 2 * y * d2.y;
    dx += -4 * x * d2.z; dy += -4 * y * d2.z;
    v4 o = {dr, dx, dy, dz};
    return o;
}

/* ---- generalized Gaussian kernel ------------------------------------------------
 * models/gaussianParticles.cuh:267-308 particleResponse, :223-265 particleResponseGrd
 * (identical in threedgrt_tracer/include/3dgrt/kernels/cuda/gaussianParticles.cuh). */
static inline real particle_response(int degree, real grayDist) {
    switch (degree) {
    case 8: { const real s = R_(-0.000685871056241); const real g2 = grayDist * grayDist; return r_exp(s * g2 * g2); }
    case 5: { const real s = R_(-0.0185185185185); return r_exp(s * grayDist * grayDist * r_sqrt(grayDist)); }
    case 4: { const real s = R_(-0.0555555555556); return r_exp(s * grayDist * grayDist); }
    case 3: { const real s = R_(-0.166666666667); return r_exp(s * grayDist * r_sqrt(grayDist)); }
    case 1: { const real s = R_(-1.5); return r_exp(s * r_sqrt(grayDist)); }
    case 0: { const real s = R_(-0.329630334487); return r_max(1 + s * r_sqrt(grayDist), 0); }
    default: { const real s = R_(-0.5); return r_exp(s * grayDist); }
    }
}
static inline real particle_response_grd(int degree, real grayDist, real gres, real gresGrd) {
    switch (degree) {
    case 8: { const real s = R_(-0.000685871056241) * R_(4); const real g2 = grayDist * grayDist; return s * g2 * grayDist * gres * gresGrd; }
    case 5: { const real s = R_(-0.0185185185185) * R_(2.5); return s * grayDist * r_sqrt(grayDist) * gres * gresGrd; }
    case 4: { const real s = R_(-0.0555555555556) * R_(2); return s * grayDist * gres * gresGrd; }
    case 3: { const real s = R_(-0.166666666667) * R_(1.5); return s * r_sqrt(grayDist) * gres * gresGrd; }
    case 1: { const real s = R_(-1.5) * R_(0.5); return s * r_sqrt(grayDist) * gres * gresGrd; }
    case 0: { const real s = R_(-0.329630334487); return gres > 0 ? (R_(0.5) * s / r_sqrt(grayDist)) * gresGrd : 0; }
    default: { const real s = R_(-0.5); return s * gres * gresGrd; }
    }
}

/* ---- spherical harmonics -----------------------------------------------------------
 * models/gaussianParticles.cuh:61-100 radianceFromSpH == sphericalHarmonics.slang:21-64 decode. */
static const double ORC_SH_C0 = 0.28209479177387814, ORC_SH_C1 = 0.4886025119029199;
static const double ORC_SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396};
static const double ORC_SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                                    -0.4570457994644658, 1.445305721320277, -0.5900435899266435};

/* basis values b[0..15] for direction d; entries above (deg+1)^2 are zero */
static inline void sh_basis(int deg, v3 d, real b[16]) {
    const real x = d.x, y = d.y, z = d.z;
    for (int i = 0; i < 16; ++i) b[i] = 0;
    b[0] = R_(ORC_SH_C0);
    if (deg > 0) {
        b[1] = -R_(ORC_SH_C1) * y; b[2] = R_(ORC_SH_C1) * z; b[3] = -R_(ORC_SH_C1) * x;
        if (deg > 1) {
            const real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = R_(ORC_SH_C2[0]) * xy; b[5] = R_(ORC_SH_C2[1]) * yz; b[6] = R_(ORC_SH_C2[2]) * (R_(2) * zz - xx - yy);
            b[7] = R_(ORC_SH_C2[3]) * xz; b[8] = R_(ORC_SH_C2[4]) * (xx - yy);
            if (deg > 2) {
                b[9]  = R_(ORC_SH_C3[0]) * y * (R_(3) * xx - yy);
                b[10] = R_(ORC_SH_C3[1]) * xy * z;
                b[11] = R_(ORC_SH_C3[2]) * y * (R_(4) * zz - xx - yy);
                b[12] = R_(ORC_SH_C3[3]) * z * (R_(2) * zz - R_(3) * xx - R_(3) * yy);
                b[13] = R_(ORC_SH_C3[4]) * x * (R_(4) * zz - xx - yy);
                b[14] = R_(ORC_SH_C3[5]) * z * (xx - yy);
                b[15] = R_(ORC_SH_C3[6]) * x * (xx - R_(3) * yy);
            }
        }
    }
}
/* d b[k] / d dir (analytic derivative of the basis; Slang autodiff of decode, sphericalHarmonics.slang:21-64) */
static inline void sh_basis_grad(int deg, v3 d, v3 g[16]) {
    const real x = d.x, y = d.y, z = d.z;
    for (int i = 0; i < 16; ++i) g[i] = v3_make(0, 0, 0);
    if (deg > 0) {
        const real c1 = R_(ORC_SH_C1);
        g[1] = v3_make(0, -c1, 0); g[2] = v3_make(0, 0, c1); g[3] = v3_make(-c1, 0, 0);
        if (deg > 1) {
            const real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            g[4] = v3_scale(v3_make(y, x, 0), R_(ORC_SH_C2[0]));
            g[5] = v3_scale(v3_make(0, z, y), R_(ORC_SH_C2[1]));
            g[6] = v3_scale(v3_make(-2 * x, -2 * y, 4 * z), R_(ORC_SH_C2[2]));
            g[7] = v3_scale(v3_make(z, 0, x), R_(ORC_SH_C2[3]));
            g[8] = v3_scale(v3_make(2 * x, -2 * y, 0), R_(ORC_SH_C2[4]));
            if (deg > 2) {
                g[9]  = v3_scale(v3_make(6 * xy, 3 * xx - 3 * yy, 0), R_(ORC_SH_C3[0]));
                g[10] = v3_scale(v3_make(yz, xz, xy), R_(ORC_SH_C3[1]));
                g[11] = v3_scale(v3_make(-2 * xy, 4 * zz - xx - 3 * yy, 8 * yz), R_(ORC_SH_C3[2]));
                g[12] = v3_scale(v3_make(-6 * xz, -6 * yz, 6 * zz - 3 * xx - 3 * yy), R_(ORC_SH_C3[3]));
                g[13] = v3_scale(v3_make(4 * zz - 3 * xx - yy, -2 * xy, 8 * xz), R_(ORC_SH_C3[4]));
                g[14] = v3_scale(v3_make(2 * xz, -2 * yz, xx - yy), R_(ORC_SH_C3[5]));
                g[15] = v3_scale(v3_make(3 * xx - 3 * yy, -6 * xy, 0), R_(ORC_SH_C3[6]));
            }
        }
    }
}
/* unclamped radiance: sum_k b_k c_k + 0.5.  coeffs: [ncoef][3] coefficient-major (models/gaussianParticles.cuh:208-221) */
static inline v3 sh_radiance_unclamped(int deg, const real* coeffs, v3 dir) {
    real b[16];
    sh_basis(deg, dir, b);
    const int n = (deg + 1) * (deg + 1);
    v3 rad = v3_make(0, 0, 0);
    for (int k = 0; k < n; ++k) {
        rad.x += b[k] * coeffs[3 * k + 0];
        rad.y += b[k] * coeffs[3 * k + 1];
        rad.z += b[k] * coeffs[3 * k + 2];
    }
    rad.x += R_(0.5); rad.y += R_(0.5); rad.z += R_(0.5);
    return rad;
}

/* ---- tiny-cuda-nn restatements (absent dependency, see file header) ---------------- */
typedef struct { real x, y, z, w; } quat_xyzw;
/* tcnn::to_mat3(quat): standard rotation, returned here as rows */
static inline m33 quat_xyzw_to_R(quat_xyzw q) {
    const real xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
    const real xz = q.x * q.z, xy = q.x * q.y, yz = q.y * q.z;
    const real wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    m33 m;
    m.r[0] = v3_make(1 - 2 * (yy + zz), 2 * (xy - wz), 2 * (xz + wy));
    m.r[1] = v3_make(2 * (xy + wz), 1 - 2 * (xx + zz), 2 * (yz - wx));
    m.r[2] = v3_make(2 * (xz - wy), 2 * (yz + wx), 1 - 2 * (xx + yy));
    return m;
}
static inline v3 m33_apply(const m33* R, v3 p) { return v3_make(v3_dot(R->r[0], p), v3_dot(R->r[1], p), v3_dot(R->r[2], p)); }
static inline m33 m33_transpose(const m33* m) {
    m33 t;
    t.r[0] = v3_make(m->r[0].x, m->r[1].x, m->r[2].x);
    t.r[1] = v3_make(m->r[0].y, m->r[1].y, m->r[2].y);
    t.r[2] = v3_make(m->r[0].z, m->r[1].z, m->r[2].z);
    return t;
}
/* tcnn::quat(mat3) == glm::quat_cast.  R given as rows: R.r[i].{x,y,z} = R[i][0..2] */
static inline quat_xyzw R_to_quat_xyzw(const m33* R) {
    const real m00 = R->r[0].x, m11 = R->r[1].y, m22 = R->r[2].z;
    const real fx = m00 - m11 - m22, fy = m11 - m00 - m22, fz = m22 - m00 - m11, fw = m00 + m11 + m22;
    int big = 0; real fb = fw;
    if (fx > fb) { fb = fx; big = 1; }
    if (fy > fb) { fb = fy; big = 2; }
    if (fz > fb) { fb = fz; big = 3; }
    const real bv = r_sqrt(fb + 1) * R_(0.5), mult = R_(0.25) / bv;
    /* element (row i, col j) */
    const real r01 = R->r[0].y, r02 = R->r[0].z, r10 = R->r[1].x, r12 = R->r[1].z, r20 = R->r[2].x, r21 = R->r[2].y;
    quat_xyzw q;
    switch (big) {
    case 0: q.w = bv; q.x = (r21 - r12) * mult; q.y = (r02 - r20) * mult; q.z = (r10 - r01) * mult; break;
    case 1: q.w = (r21 - r12) * mult; q.x = bv; q.y = (r10 + r01) * mult; q.z = (r02 + r20) * mult; break;
    case 2: q.w = (r02 - r20) * mult; q.x = (r10 + r01) * mult; q.y = bv; q.z = (r21 + r12) * mult; break;
    default: q.w = (r10 - r01) * mult; q.x = (r02 + r20) * mult; q.y = (r21 + r12) * mult; q.z = bv; break;
    }
    return q;
}
/* tcnn::slerp (glm::slerp) */
static inline quat_xyzw quat_slerp(quat_xyzw a, quat_xyzw b, real t) {
    real c = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    if (c < 0) { b.x = -b.x; b.y = -b.y; b.z = -b.z; b.w = -b.w; c = -c; }
    const real eps = sizeof(real) == 4 ? R_(1.1920929e-07) : R_(2.220446049250313e-16);
    quat_xyzw o;
    if (c > 1 - eps) {
        o.x = a.x * (1 - t) + b.x * t; o.y = a.y * (1 - t) + b.y * t; o.z = a.z * (1 - t) + b.z * t; o.w = a.w * (1 - t) + b.w * t;
    } else {
        const real ang = r_acos(c), s0 = r_sin((1 - t) * ang), s1 = r_sin(t * ang), is = 1 / r_sin(ang);
        o.x = (s0 * a.x + s1 * b.x) * is; o.y = (s0 * a.y + s1 * b.y) * is; o.z = (s0 * a.z + s1 * b.z) * is; o.w = (s0 * a.w + s1 * b.w) * is;
    }
    return o;
}

/* sensor pose [t(3), q(x,y,z,w)] (sensors/sensors.h:33) */
typedef struct { v3 t; quat_xyzw q; } orc_pose;
static inline orc_pose pose_from7(const real* p) {
    orc_pose o; o.t = v3_make(p[0], p[1], p[2]); o.q.x = p[3]; o.q.y = p[4]; o.q.z = p[5]; o.q.w = p[6]; return o;
}
/* sensors.h:53-66 interpolatedSensorPose */
static inline orc_pose pose_interpolate(orc_pose a, orc_pose b, real t) {
    orc_pose o;
    o.q = quat_slerp(a.q, b.q, t);
    o.t = v3_add(v3_scale(a.t, 1 - t), v3_scale(b.t, t));
    return o;
}
/* sensors.h:44-51 sensorPoseInverse */
static inline orc_pose pose_inverse(orc_pose p) {
    const m33 R = quat_xyzw_to_R(p.q);
    const m33 Rt = m33_transpose(&R);
    orc_pose o;
    o.q = R_to_quat_xyzw(&Rt);
    o.t = v3_scale(m33_apply(&Rt, p.t), -1);
    return o;
}

#endif /* ORC_MATH_H */
