// cuda_shim.h — lets the reference's hand-written per-hit CUDA headers compile with g++ on the host.
// TEST INFRASTRUCTURE ONLY (oracle/_ref): it defines the handful of CUDA built-ins those headers use;
// it contains no reference code.  See SURVEY.md Appendix B.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#define __CUDACC__ 1
#define __device__
#define __host__
#define __forceinline__ inline
#define __inline__ inline
#define __restrict__
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline int3 make_int3(int x, int y, int z) { return {x, y, z}; }
inline int2 make_int2(int x, int y) { return {x, y}; }
inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { return {x, y, z}; }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
inline float atomicAdd(float* a, float v) { float o = *a; *a += v; return o; }
template <class T> inline T atomicMin(T* a, T v) { T o = *a; *a = std::min(o, v); return o; }
template <class T> inline T atomicMax(T* a, T v) { T o = *a; *a = std::max(o, v); return o; }
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline unsigned __float_as_uint(float f) { unsigned i; std::memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline float __uint_as_float(unsigned i) { float f; std::memcpy(&f, &i, 4); return f; }
using std::max;
using std::min;
