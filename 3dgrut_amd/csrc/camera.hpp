// camera.hpp — sensor poses (host) and camera projections (device) for the 3DGUT projector.
// Behaviour follows threedgut_tracer/include/3dgut/sensors/sensors.h:44-73 and
// kernels/cuda/sensors/cameraProjections.cuh:24-257; tiny-cuda-nn's quat/mat helpers (un-vendored
// submodule) are restated from their published glm-compatible definitions.
#pragma once

#include <cfloat>
#include <cmath>

#include "common.hpp"

namespace grut {

// world->sensor pose [t, q(x,y,z,w)]
struct Pose {
    float t[3];
    float q[4];  // x,y,z,w
};

// rotation matrix rows from an (x,y,z,w) quaternion (tcnn::to_mat3)
__host__ __device__ inline void quat_xyzw_to_rows(const float q[4], float R[9]) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.f - 2.f * (yy + zz); R[1] = 2.f * (xy - wz); R[2] = 2.f * (xz + wy);
    R[3] = 2.f * (xy + wz); R[4] = 1.f - 2.f * (xx + zz); R[5] = 2.f * (yz - wx);
    R[6] = 2.f * (xz - wy); R[7] = 2.f * (yz + wx); R[8] = 1.f - 2.f * (xx + yy);
}

// sinf / acosf of the pose interpolation, evaluated in float64 and rounded once: the reference interpolates its poses on the HOST
// (sensors.h:53-66 through glibc, whose float functions are correctly rounded in all but ~1e-8 of the cases); the device's own
// sinf / acosf round differently, and at a slerp angle of ~5e-4 - the "angle" between a unit quaternion and itself when its float32
// norm is a few ulp below one - the quotient sin(t a) / sin(a) passes such a difference on to every entry of the view matrix.
__host__ __device__ inline float pose_sinf(float x) { return (float)sin((double)x); }
__host__ __device__ inline float pose_acosf(float x) { return (float)acos((double)x); }

// tcnn::slerp (glm::slerp)
__host__ __device__ inline void quat_slerp(const float a[4], const float b_in[4], float t, float o[4]) {
    float b[4] = {b_in[0], b_in[1], b_in[2], b_in[3]};
    float c = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    if (c < 0.f) {
        for (int i = 0; i < 4; ++i) b[i] = -b[i];
        c = -c;
    }
    if (c > 1.f - FLT_EPSILON) {
        for (int i = 0; i < 4; ++i) o[i] = a[i] * (1.f - t) + b[i] * t;
    } else {
        const float ang = pose_acosf(c), s0 = pose_sinf((1.f - t) * ang), s1 = pose_sinf(t * ang), is = 1.f / pose_sinf(ang);
        for (int i = 0; i < 4; ++i) o[i] = (s0 * a[i] + s1 * b[i]) * is;
    }
}

// tcnn::quat(mat3) == glm::quat_cast; R as rows
__host__ __device__ inline void rows_to_quat_xyzw(const float R[9], float q[4]) {
    const float m00 = R[0], m11 = R[4], m22 = R[8];
    const float fx = m00 - m11 - m22, fy = m11 - m00 - m22, fz = m22 - m00 - m11, fw = m00 + m11 + m22;
    int big = 0;
    float fb = fw;
    if (fx > fb) { fb = fx; big = 1; }
    if (fy > fb) { fb = fy; big = 2; }
    if (fz > fb) { fb = fz; big = 3; }
    const float bv = sqrtf(fb + 1.f) * 0.5f, mult = 0.25f / bv;
    const float r01 = R[1], r02 = R[2], r10 = R[3], r12 = R[5], r20 = R[6], r21 = R[7];
    switch (big) {
    case 0: q[3] = bv; q[0] = (r21 - r12) * mult; q[1] = (r02 - r20) * mult; q[2] = (r10 - r01) * mult; break;
    case 1: q[3] = (r21 - r12) * mult; q[0] = bv; q[1] = (r10 + r01) * mult; q[2] = (r02 + r20) * mult; break;
    case 2: q[3] = (r02 - r20) * mult; q[0] = (r10 + r01) * mult; q[1] = bv; q[2] = (r21 + r12) * mult; break;
    default: q[3] = (r10 - r01) * mult; q[0] = (r02 + r20) * mult; q[1] = (r21 + r12) * mult; q[2] = bv; break;
    }
}

__host__ __device__ inline Pose pose_interpolate(const Pose& a, const Pose& b, float t) {  // sensors.h:53-66
    Pose o;
    quat_slerp(a.q, b.q, t, o.q);
    for (int i = 0; i < 3; ++i) o.t[i] = a.t[i] * (1.f - t) + b.t[i] * t;
    return o;
}
__host__ __device__ inline Pose pose_inverse(const Pose& p) {  // sensors.h:44-51
    float R[9], Rt[9];
    quat_xyzw_to_rows(p.q, R);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rt[3 * i + j] = R[3 * j + i];
    Pose o;
    rows_to_quat_xyzw(Rt, o.q);
    for (int i = 0; i < 3; ++i) o.t[i] = -(Rt[3 * i] * p.t[0] + Rt[3 * i + 1] * p.t[1] + Rt[3 * i + 2] * p.t[2]);
    return o;
}

// everything the kernels need about the frame's poses
struct FramePoses {
    float start_R[9], start_t[3], start_q[4];
    float end_t[3], end_q[4];
    float view_R[9], view_t[3];  // mid-exposure world->sensor
    float s2w_R[9], s2w_t[3];    // sensor->world of the mid-exposure pose (gutRenderer.cu:266-267, 407)
};

__host__ __device__ inline FramePoses make_frame_poses(const float ps7[7], const float pe7[7]) {
    Pose s, e;
    for (int i = 0; i < 3; ++i) { s.t[i] = ps7[i]; e.t[i] = pe7[i]; }
    for (int i = 0; i < 4; ++i) { s.q[i] = ps7[3 + i]; e.q[i] = pe7[3 + i]; }
    const Pose mid = pose_interpolate(s, e, 0.5f);
    const Pose inv = pose_inverse(mid);
    FramePoses f;
    quat_xyzw_to_rows(s.q, f.start_R);
    for (int i = 0; i < 3; ++i) { f.start_t[i] = s.t[i]; f.end_t[i] = e.t[i]; f.view_t[i] = mid.t[i]; f.s2w_t[i] = inv.t[i]; }
    for (int i = 0; i < 4; ++i) { f.start_q[i] = s.q[i]; f.end_q[i] = e.q[i]; }
    quat_xyzw_to_rows(mid.q, f.view_R);
    quat_xyzw_to_rows(inv.q, f.s2w_R);
    return f;
}

// [t, q(x,y,z,w)] of the inverse of a camera-to-world matrix (row-major 4x4, rows 0-2 are read): the arithmetic of the reference
// plugin's host code, step by step, for poses that live in device memory —
//   (1) threedgut_tracer/tracer.py:413-419: the float32 pose is widened to FLOAT64 (np.concatenate with a float64 row promotes), row 3
//       is set to (0,0,0,1) and the GENERAL 4x4 inverse is taken with np.linalg.inv (LAPACK dgesv: LU with partial pivoting, solve
//       against the identity) — not a rigid R^T: a float32 rotation is orthogonal to ~6e-8 only, and the general inverse differs from the
//       transpose at exactly that level;
//   (2) tracer.py:366-367: ONE rounding of t and R to float32;
//   (3) tracer.py:88-136 (SensorPose3DModel.__so3_matrix_to_quat): the quaternion from the ROUNDED matrix in float32, operation by
//       operation (decision sums left to right, `(1 - tr) + 2 r_ii`, squared norm summed left to right, one sqrt, one division each).
// LAPACK's float64 BITS depend on the BLAS build; after the rounding of (2) every LU of that kind gives the same float32 values except
// where an entry is a cancellation residue (|x| < 1e-9, absolute difference < 1e-15): tests/test_host_cpu.py pins this function against
// tests/golden/pose.npz (produced by the reference's own Python), bit for bit outside that class.
// Must be compiled WITHOUT floating-point contraction (csrc/gut_poses.hip, -ffp-contract=off).
// LU with partial pivoting of [rows 0-2 of m | 0 0 0 1] (dgetf2: pivot = largest magnitude of the column, scale by the reciprocal,
// rank-1 update); perm[i] = original row now in position i
struct PoseLU {
    double A[4][4];
    int perm[4];
};
// (every index below is a compile-time constant after unrolling - row exchanges and the quaternion's branch are written with selects /
// a switch - so that the device keeps the matrices in registers: one dynamically indexed access would put them in scratch memory, a
// dependent ~1 us round trip each; measured 22 us for this kernel against 5)
__host__ __device__ inline PoseLU pose_lu_factor(const float* m) {
    PoseLU f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f.perm[i] = i;
#pragma unroll
        for (int j = 0; j < 4; ++j) f.A[i][j] = i < 3 ? (double)m[4 * i + j] : (j == 3 ? 1.0 : 0.0);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int p = k;
        double best = fabs(f.A[k][k]);
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            const double v = fabs(f.A[i][k]);
            if (v > best) { best = v; p = i; }
        }
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {   // exchange rows k and p
            const bool sw = p == i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double a = f.A[k][j], b = f.A[i][j];
                f.A[k][j] = sw ? b : a;
                f.A[i][j] = sw ? a : b;
            }
            const int pa = f.perm[k], pb = f.perm[i];
            f.perm[k] = sw ? pb : pa;
            f.perm[i] = sw ? pa : pb;
        }
        const double r = 1.0 / f.A[k][k];
#pragma unroll
        for (int i = k + 1; i < 4; ++i) f.A[i][k] = f.A[i][k] * r;
#pragma unroll
        for (int i = k + 1; i < 4; ++i)
#pragma unroll
            for (int j = k + 1; j < 4; ++j) f.A[i][j] = f.A[i][j] - f.A[i][k] * f.A[k][j];
    }
    return f;
}
// column c of the inverse (dgetrs against the permuted identity: unit-lower forward substitution, upper back substitution)
__host__ __device__ inline void pose_lu_solve_column(const PoseLU& f, int c, double x[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = f.perm[i] == c ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) x[i] = x[i] - f.A[i][j] * x[j];
#pragma unroll
    for (int i = 3; i >= 0; --i) {
#pragma unroll
        for (int j = i + 1; j < 4; ++j) x[i] = x[i] - f.A[i][j] * x[j];
        x[i] = x[i] / f.A[i][i];
    }
}
// steps (2) and (3): [t, q] from the float32-rounded rows {R | t} of the inverse
__host__ __device__ inline void pose_tquat_from_rows(const float R[9], const float t[3], float out7[7]) {
    out7[0] = t[0]; out7[1] = t[1]; out7[2] = t[2];
    const float d0 = R[0], d1 = R[4], d2 = R[8], d3 = (d0 + d1) + d2;
    int c = 0;
    float best = d0;
    if (d1 > best) { best = d1; c = 1; }
    if (d2 > best) { best = d2; c = 2; }
    if (d3 > best) { best = d3; c = 3; }
    float q0, q1, q2, q3;
    switch (c) {   // i = c, j = (c + 1) % 3, k = (c + 2) % 3 of tracer.py:117-124, spelled out
    case 0: q0 = (1.f - d3) + 2.f * R[0]; q1 = R[3] + R[1]; q2 = R[6] + R[2]; q3 = R[7] - R[5]; break;
    case 1: q1 = (1.f - d3) + 2.f * R[4]; q2 = R[7] + R[5]; q0 = R[1] + R[3]; q3 = R[2] - R[6]; break;
    case 2: q2 = (1.f - d3) + 2.f * R[8]; q0 = R[2] + R[6]; q1 = R[5] + R[7]; q3 = R[3] - R[1]; break;
    default: q0 = R[7] - R[5]; q1 = R[2] - R[6]; q2 = R[3] - R[1]; q3 = 1.f + d3; break;
    }
    const float n2 = ((q0 * q0 + q1 * q1) + q2 * q2) + q3 * q3;
    // correctly rounded float sqrt and quotients through float64 (53 >= 2*24 + 2 bits: no double rounding), whatever the
    // compiler's float sqrt / division expansion is
    const double n = (double)(float)sqrt((double)n2);
    out7[3] = (float)((double)q0 / n); out7[4] = (float)((double)q1 / n); out7[5] = (float)((double)q2 / n); out7[6] = (float)((double)q3 / n);
}
__host__ __device__ inline void c2w_to_world_to_sensor(const float* m, float out7[7]) {
    const PoseLU f = pose_lu_factor(m);
    float R[9], t[3];
    for (int c = 0; c < 4; ++c) {
        double x[4];
        pose_lu_solve_column(f, c, x);
        for (int i = 0; i < 3; ++i) {
            if (c < 3) R[3 * i + c] = (float)x[i];
            else t[i] = (float)x[i];
        }
    }
    pose_tquat_from_rows(R, t, out7);
}

#ifdef __HIPCC__
// ---- projections (device) -------------------------------------------------------------------
__device__ __forceinline__ bool within_resolution(float w, float h, float tol, float px, float py) {
    const float mx = w * tol, my = h * tol;
    return (px > -mx) && (py > -my) && (px < w + mx) && (py < h + my);
}
__device__ __forceinline__ float stable_norm2(float x, float y) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mn = fminf(ax, ay), mx = fmaxf(ax, ay);
    if (mx <= 0.f) return 0.f;
    const float r = mn / mx;
    return mx * sqrtf(1.f + r * r);
}
template <int N>
__device__ __forceinline__ float poly_horner(const float* c, float x) {
    float y = c[N - 1];
#pragma unroll
    for (int i = N - 2; i >= 0; --i) y = x * y + c[i];
    return y;
}

__device__ inline bool project_pinhole(const GrutCamera& cam, f3 p, float tol, float& ox, float& oy) {
    if (p.z <= 0.f) { ox = 0.f; oy = 0.f; return false; }
    const float iz = 1.f / p.z;
    const float u = p.x * iz, v = p.y * iz;
    const float u2 = u * u, v2 = v * v, r2 = u2 + v2;
    const float a1 = 2.f * u * v, a2 = r2 + 2.f * u2, a3 = r2 + 2.f * v2;
    const float num = 1.f + r2 * (cam.radial[0] + r2 * (cam.radial[1] + r2 * cam.radial[2]));
    const float den = 1.f + r2 * (cam.radial[3] + r2 * (cam.radial[4] + r2 * cam.radial[5]));
    const float icD = num / den;
    const float dx = cam.tangential[0] * a1 + cam.tangential[1] * a2 + r2 * (cam.thin_prism[0] + r2 * cam.thin_prism[1]);
    const float dy = cam.tangential[0] * a3 + cam.tangential[1] * a1 + r2 * (cam.thin_prism[2] + r2 * cam.thin_prism[3]);
    const bool valid_radial = (icD > 0.8f) && (icD < 1.2f);
    if (valid_radial) {
        ox = (icD * u + dx) * cam.focal_length[0] + cam.principal_point[0];
        oy = (icD * v + dy) * cam.focal_length[1] + cam.principal_point[1];
    } else {
        const float clip = hypotf((float)cam.width, (float)cam.height);
        const float s = clip / sqrtf(r2);
        ox = s * u + cam.principal_point[0];
        oy = s * v + cam.principal_point[1];
    }
    return valid_radial && within_resolution((float)cam.width, (float)cam.height, tol, ox, oy);
}
__device__ inline bool project_fisheye(const GrutCamera& cam, f3 p, float tol, float& ox, float& oy) {
    float rho = stable_norm2(p.x, p.y);
    if (rho <= 0.f) rho = FLT_EPSILON;
    const float theta_full = atan2f(rho, p.z);
    const float theta = fminf(theta_full, cam.max_angle);
    const float t2 = theta * theta;
    const float delta = (theta * (poly_horner<4>(cam.radial, t2) * t2 + 1.f)) / rho;
    ox = cam.focal_length[0] * p.x * delta + cam.principal_point[0];
    oy = cam.focal_length[1] * p.y * delta + cam.principal_point[1];
    return (theta < cam.max_angle) && within_resolution((float)cam.width, (float)cam.height, tol, ox, oy);
}
__device__ inline bool project_ftheta(const GrutCamera& cam, f3 p, float tol, float& ox, float& oy) {
    float rho = stable_norm2(p.x, p.y);
    if (rho <= 0.f) rho = FLT_EPSILON;
    const float theta_full = atan2f(rho, p.z);
    const float theta = fminf(theta_full, cam.max_angle);
    float delta = poly_horner<6>(cam.ftheta_angle_to_pixeldist, theta);
    if (cam.ftheta_reference_poly == GRUT_FTHETA_PIXELDIST_TO_ANGLE) {
        float dpoly[5];
#pragma unroll
        for (int i = 1; i < 6; ++i) dpoly[i - 1] = (float)i * cam.ftheta_pixeldist_to_angle[i];
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const float dfdx = poly_horner<5>(dpoly, delta);
            const float res  = poly_horner<6>(cam.ftheta_pixeldist_to_angle, delta) - theta;
            delta -= res / dfdx;
        }
    }
    const float s = delta / rho;
    ox = s * (cam.ftheta_linear_cde[0] * p.x + cam.ftheta_linear_cde[1] * p.y) + cam.principal_point[0] + 0.5f;
    oy = s * (cam.ftheta_linear_cde[2] * p.x + p.y) + cam.principal_point[1] + 0.5f;
    return (theta < cam.max_angle) && within_resolution((float)cam.width, (float)cam.height, tol, ox, oy);
}
// MODEL: the camera model when the caller knows it at compile time (kernels specialised per model), -1: decided at run time
template <int MODEL = -1>
__device__ __forceinline__ bool project_point(const GrutCamera& cam, f3 p, float tol, float& ox, float& oy) {
    switch (MODEL >= 0 ? MODEL : cam.model) {
    case GRUT_CAMERA_OPENCV_PINHOLE: return project_pinhole(cam, p, tol, ox, oy);
    case GRUT_CAMERA_OPENCV_FISHEYE: return project_fisheye(cam, p, tol, ox, oy);
    case GRUT_CAMERA_FTHETA: return project_ftheta(cam, p, tol, ox, oy);
    default: ox = 0.f; oy = 0.f; return false;
    }
}
__device__ __forceinline__ f3 apply_rows(const float R[9], const float t[3], f3 p) {
    return f3{fmaf(R[0], p.x, fmaf(R[1], p.y, fmaf(R[2], p.z, t[0]))), fmaf(R[3], p.x, fmaf(R[4], p.y, fmaf(R[5], p.z, t[1]))),
              fmaf(R[6], p.x, fmaf(R[7], p.y, fmaf(R[8], p.z, t[2])))};
}
__device__ __forceinline__ float relative_shutter_time(const GrutCamera& cam, float px, float py) {
    switch (cam.shutter) {
    case GRUT_SHUTTER_ROLLING_TOP_TO_BOTTOM: return floorf(py) / ((float)cam.height - 1.f);
    case GRUT_SHUTTER_ROLLING_LEFT_TO_RIGHT: return floorf(px) / ((float)cam.width - 1.f);
    case GRUT_SHUTTER_ROLLING_BOTTOM_TO_TOP: return ((float)cam.height - ceilf(py)) / ((float)cam.height - 1.f);
    case GRUT_SHUTTER_ROLLING_RIGHT_TO_LEFT: return ((float)cam.width - ceilf(px)) / ((float)cam.width - 1.f);
    default: return 0.5f;
    }
}
// cameraProjections.cuh:218-257
// ROLLING: 0 = the frame has a global shutter (compile-time knowledge), 1 = decided at run time
template <int MODEL = -1, int ROLLING = 1>
__device__ inline bool project_point_with_shutter(const GrutCamera& cam, const FramePoses& fp, int n_iter, f3 x, float tol,
                                                  float& ox, float& oy) {
    bool valid = project_point<MODEL>(cam, apply_rows(fp.start_R, fp.start_t, x), tol, ox, oy);
    if (!ROLLING || cam.shutter == GRUT_SHUTTER_GLOBAL) return valid;
    if (!valid) {
        float Re[9];
        quat_xyzw_to_rows(fp.end_q, Re);
        valid = project_point<MODEL>(cam, apply_rows(Re, fp.end_t, x), tol, ox, oy);
        if (!valid) return false;
    }
    for (int i = 0; i < n_iter; ++i) {
        const float a = relative_shutter_time(cam, ox, oy);
        float q[4], R[9], t[3];
        quat_slerp(fp.start_q, fp.end_q, a, q);
        quat_xyzw_to_rows(q, R);
        for (int k = 0; k < 3; ++k) t[k] = fp.start_t[k] * (1.f - a) + fp.end_t[k] * a;
        valid = project_point<MODEL>(cam, apply_rows(R, t, x), tol, ox, oy);
    }
    return valid;
}
#endif  // __HIPCC__

}  // namespace grut
