// Host stand-in for the part of tiny-cuda-nn's vec.h that the reference's sensor / camera headers use
// (threedgut_tracer/include/3dgut/sensors/*.h, kernels/cuda/sensors/cameraProjections.cuh).  tiny-cuda-nn is an
// un-vendored submodule of the reference (empty directory, no pinned SHA): these few types restate its GLM-compatible
// definitions — column-major matrices (m[i] = i-th column), quaternions constructed as (w, x, y, z), GLM's mat3_cast /
// quat_cast / slerp / mix.  TEST INFRASTRUCTURE ONLY (oracle/_ref); contains no reference code.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <type_traits>

#ifndef TCNN_HOST_DEVICE
#define TCNN_HOST_DEVICE
#endif

namespace tcnn {

template <typename T, uint32_t N, size_t A = sizeof(T)>
struct tvec;

// element access / slicing shared by all sizes (the structs below are contiguous arrays of T)
#define TCNN_SHIM_VEC_COMMON(N_)                                                                         \
    T* data() { return reinterpret_cast<T*>(this); }                                                     \
    const T* data() const { return reinterpret_cast<const T*>(this); }                                   \
    T& operator[](uint32_t i) { return data()[i]; }                                                      \
    const T& operator[](uint32_t i) const { return data()[i]; }                                          \
    template <uint32_t O, uint32_t S>                                                                    \
    tvec<T, S, A>& slice() { return *reinterpret_cast<tvec<T, S, A>*>(data() + O); }                     \
    template <uint32_t O, uint32_t S>                                                                    \
    const tvec<T, S, A>& slice() const { return *reinterpret_cast<const tvec<T, S, A>*>(data() + O); }   \
    static tvec zero() { tvec r; for (uint32_t i = 0; i < N_; ++i) r[i] = T(0); return r; }              \
    /* converts the element type and, like tcnn, truncates a longer vector to its leading components */  \
    template <typename U, uint32_t M_, size_t B, typename = typename std::enable_if<(M_ >= N_)>::type>   \
    tvec(const tvec<U, M_, B>& o) { for (uint32_t i = 0; i < N_; ++i) (*this)[i] = (T)o[i]; }            \
    tvec& operator+=(const tvec& o) { for (uint32_t i = 0; i < N_; ++i) (*this)[i] += o[i]; return *this; } \
    tvec& operator-=(const tvec& o) { for (uint32_t i = 0; i < N_; ++i) (*this)[i] -= o[i]; return *this; } \
    tvec& operator*=(T s) { for (uint32_t i = 0; i < N_; ++i) (*this)[i] *= s; return *this; }           \
    tvec() = default;

template <typename T, uint32_t N, size_t A>
struct tvec {
    T v[N];
    TCNN_SHIM_VEC_COMMON(N)
};
template <typename T, size_t A>
struct tvec<T, 2, A> {
    T x, y;
    tvec(T x_, T y_) : x(x_), y(y_) {}
    explicit tvec(T s) : x(s), y(s) {}
    template <typename U, typename V, typename = typename std::enable_if<!std::is_same<U, T>::value || !std::is_same<V, T>::value>::type>
    tvec(U x_, V y_) : x((T)x_), y((T)y_) {}
    TCNN_SHIM_VEC_COMMON(2)
};
template <typename T, size_t A>
struct tvec<T, 3, A> {
    T x, y, z;
    tvec(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
    explicit tvec(T s) : x(s), y(s), z(s) {}
    tvec<T, 2, A> xy() const { return tvec<T, 2, A>(x, y); }
    TCNN_SHIM_VEC_COMMON(3)
};
template <typename T, size_t A>
struct tvec<T, 4, A> {
    T x, y, z, w;
    tvec(T x_, T y_, T z_, T w_) : x(x_), y(y_), z(z_), w(w_) {}
    explicit tvec(T s) : x(s), y(s), z(s), w(s) {}
    tvec(const tvec<T, 3, A>& v, T w_) : x(v.x), y(v.y), z(v.z), w(w_) {}   // GLM vec4(vec3, w)
    TCNN_SHIM_VEC_COMMON(4)
};

template <uint32_t N>
using vec = tvec<float, N>;
using vec2 = vec<2>;
using vec3 = vec<3>;
using vec4 = vec<4>;
using ivec2 = tvec<int, 2>;
using uvec2 = tvec<uint32_t, 2>;

#define TCNN_SHIM_VEC_OP(op)                                                                                  \
    template <typename T, uint32_t N, size_t A>                                                               \
    tvec<T, N, A> operator op(const tvec<T, N, A>& a, const tvec<T, N, A>& b) {                               \
        tvec<T, N, A> r; for (uint32_t i = 0; i < N; ++i) r[i] = a[i] op b[i]; return r; }                    \
    template <typename T, uint32_t N, size_t A>                                                               \
    tvec<T, N, A> operator op(const tvec<T, N, A>& a, T b) {                                                  \
        tvec<T, N, A> r; for (uint32_t i = 0; i < N; ++i) r[i] = a[i] op b; return r; }                       \
    template <typename T, uint32_t N, size_t A>                                                               \
    tvec<T, N, A> operator op(T a, const tvec<T, N, A>& b) {                                                  \
        tvec<T, N, A> r; for (uint32_t i = 0; i < N; ++i) r[i] = a op b[i]; return r; }
TCNN_SHIM_VEC_OP(+)
TCNN_SHIM_VEC_OP(-)
TCNN_SHIM_VEC_OP(*)
TCNN_SHIM_VEC_OP(/)
#undef TCNN_SHIM_VEC_OP

template <typename T, uint32_t N, size_t A>
T dot(const tvec<T, N, A>& a, const tvec<T, N, A>& b) { T s = T(0); for (uint32_t i = 0; i < N; ++i) s += a[i] * b[i]; return s; }
template <typename T, uint32_t N, size_t A>
T length(const tvec<T, N, A>& a) { return std::sqrt(dot(a, a)); }
template <typename T, uint32_t N, size_t A>
tvec<T, N, A> normalize(const tvec<T, N, A>& a) { return a / length(a); }
#define TCNN_SHIM_VEC_FN2(name, expr)                                                     \
    template <typename T, uint32_t N, size_t A>                                           \
    tvec<T, N, A> name(const tvec<T, N, A>& a, const tvec<T, N, A>& b) {                  \
        tvec<T, N, A> r; for (uint32_t i = 0; i < N; ++i) r[i] = (expr); return r; }
TCNN_SHIM_VEC_FN2(min, a[i] < b[i] ? a[i] : b[i])
TCNN_SHIM_VEC_FN2(max, a[i] > b[i] ? a[i] : b[i])
TCNN_SHIM_VEC_FN2(copysign, std::copysign(a[i], b[i]))
#undef TCNN_SHIM_VEC_FN2
template <typename T, uint32_t N, size_t A>
tvec<T, N, A> max(const tvec<T, N, A>& a, T b) { tvec<T, N, A> r; for (uint32_t i = 0; i < N; ++i) r[i] = a[i] > b ? a[i] : b; return r; }
template <typename T, uint32_t N, size_t A>
tvec<T, N, A> sqrt(const tvec<T, N, A>& a) { tvec<T, N, A> r; for (uint32_t i = 0; i < N; ++i) r[i] = std::sqrt(a[i]); return r; }
// per-component blend with a vector of weights (GLM mix(x, y, a) with vector a)
template <typename T, uint32_t N, size_t A>
tvec<T, N, A> mix(const tvec<T, N, A>& x, const tvec<T, N, A>& y, const tvec<T, N, A>& a) { return x * (tvec<T, N, A>(T(1)) - a) + y * a; }
inline float sqrt(float a) { return std::sqrt(a); }
template <typename T, uint32_t N, size_t A>
T length2(const tvec<T, N, A>& a) { return dot(a, a); }
template <typename T, uint32_t N, size_t A>
tvec<T, N, A> abs(const tvec<T, N, A>& a) { tvec<T, N, A> r; for (uint32_t i = 0; i < N; ++i) r[i] = std::abs(a[i]); return r; }
template <typename T, uint32_t N, size_t A>
T max(const tvec<T, N, A>& a) { T r = a[0]; for (uint32_t i = 1; i < N; ++i) r = a[i] > r ? a[i] : r; return r; }   // largest component
template <typename T> T div_round_up(T a, T b) { return (a + b - 1) / b; }
template <typename T> void host_device_swap(T& a, T& b) { T t = a; a = b; b = t; }
// GLM mix: x * (1 - a) + y * a
template <typename T, uint32_t N, size_t A>
tvec<T, N, A> mix(const tvec<T, N, A>& x, const tvec<T, N, A>& y, T a) { return x * (T(1) - a) + y * a; }

// column-major N columns x M rows
template <typename T, uint32_t N, uint32_t M>
struct tmat {
    tvec<T, M> m[N];
    tmat() = default;
    tmat(const tvec<T, M>& c0, const tvec<T, M>& c1, const tvec<T, M>& c2) { m[0] = c0; m[1] = c1; m[2] = c2; }
    tmat(const tvec<T, M>& c0, const tvec<T, M>& c1, const tvec<T, M>& c2, const tvec<T, M>& c3) { m[0] = c0; m[1] = c1; m[2] = c2; m[3] = c3; }
    // GLM matNxM(matPxQ): the upper-left block of a larger matrix
    template <uint32_t P, uint32_t Q, typename = typename std::enable_if<(P >= N && Q >= M && (P != N || Q != M))>::type>
    explicit tmat(const tmat<T, P, Q>& o) { for (uint32_t c = 0; c < N; ++c) for (uint32_t r = 0; r < M; ++r) m[c][r] = o[c][r]; }
    tvec<T, M>& operator[](uint32_t i) { return m[i]; }
    const tvec<T, M>& operator[](uint32_t i) const { return m[i]; }
};
using mat3 = tmat<float, 3, 3>;
using mat4x3 = tmat<float, 4, 3>;

template <typename T, uint32_t N, uint32_t M>
tvec<T, M> operator*(const tmat<T, N, M>& a, const tvec<T, N>& v) {
    tvec<T, M> r = tvec<T, M>::zero();
    for (uint32_t c = 0; c < N; ++c) r = r + a[c] * v[c];
    return r;
}
template <typename T, uint32_t N, uint32_t M>
tmat<T, N, M> operator*(T s, const tmat<T, N, M>& a) { tmat<T, N, M> r; for (uint32_t c = 0; c < N; ++c) r[c] = a[c] * s; return r; }
template <typename T, uint32_t N>
tmat<T, N, N> transpose(const tmat<T, N, N>& a) {
    tmat<T, N, N> r;
    for (uint32_t c = 0; c < N; ++c) for (uint32_t k = 0; k < N; ++k) r[c][k] = a[k][c];
    return r;
}

template <typename T>
struct tquat {
    T x, y, z, w;
    tquat() = default;
    tquat(T w_, T x_, T y_, T z_) : x(x_), y(y_), z(z_), w(w_) {}   // GLM order of arguments
    // GLM quat_cast
    explicit tquat(const tmat<T, 3, 3>& m) {
        const T fx = m[0][0] - m[1][1] - m[2][2], fy = m[1][1] - m[0][0] - m[2][2], fz = m[2][2] - m[0][0] - m[1][1];
        const T fw = m[0][0] + m[1][1] + m[2][2];
        int big = 0;
        T best = fw;
        if (fx > best) { best = fx; big = 1; }
        if (fy > best) { best = fy; big = 2; }
        if (fz > best) { best = fz; big = 3; }
        const T val = std::sqrt(best + T(1)) * T(0.5), mult = T(0.25) / val;
        switch (big) {
        case 0: w = val; x = (m[1][2] - m[2][1]) * mult; y = (m[2][0] - m[0][2]) * mult; z = (m[0][1] - m[1][0]) * mult; break;
        case 1: w = (m[1][2] - m[2][1]) * mult; x = val; y = (m[0][1] + m[1][0]) * mult; z = (m[2][0] + m[0][2]) * mult; break;
        case 2: w = (m[2][0] - m[0][2]) * mult; x = (m[0][1] + m[1][0]) * mult; y = val; z = (m[1][2] + m[2][1]) * mult; break;
        default: w = (m[0][1] - m[1][0]) * mult; x = (m[2][0] + m[0][2]) * mult; y = (m[1][2] + m[2][1]) * mult; z = val; break;
        }
    }
};
using quat = tquat<float>;

// GLM mat3_cast
template <typename T>
tmat<T, 3, 3> to_mat3(const tquat<T>& q) {
    tmat<T, 3, 3> r;
    const T qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z, qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
    const T qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    r[0][0] = T(1) - T(2) * (qyy + qzz); r[0][1] = T(2) * (qxy + qwz); r[0][2] = T(2) * (qxz - qwy);
    r[1][0] = T(2) * (qxy - qwz); r[1][1] = T(1) - T(2) * (qxx + qzz); r[1][2] = T(2) * (qyz + qwx);
    r[2][0] = T(2) * (qxz + qwy); r[2][1] = T(2) * (qyz - qwx); r[2][2] = T(1) - T(2) * (qxx + qyy);
    return r;
}
// GLM slerp: shortest path, linear interpolation when the quaternions are nearly parallel
template <typename T>
tquat<T> slerp(const tquat<T>& x, const tquat<T>& y, T a) {
    tquat<T> z = y;
    T c = x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    if (c < T(0)) { z = tquat<T>(-y.w, -y.x, -y.y, -y.z); c = -c; }
    if (c > T(1) - std::numeric_limits<T>::epsilon()) {
        return tquat<T>(x.w * (T(1) - a) + z.w * a, x.x * (T(1) - a) + z.x * a, x.y * (T(1) - a) + z.y * a, x.z * (T(1) - a) + z.z * a);
    }
    const T angle = std::acos(c);
    const T s0 = std::sin((T(1) - a) * angle), s1 = std::sin(a * angle), inv = T(1) / std::sin(angle);
    return tquat<T>((s0 * x.w + s1 * z.w) * inv, (s0 * x.x + s1 * z.x) * inv, (s0 * x.y + s1 * z.y) * inv, (s0 * x.z + s1 * z.z) * inv);
}

}  // namespace tcnn
