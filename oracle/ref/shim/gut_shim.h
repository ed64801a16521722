// gut_shim.h — the 3DGUT headers rely on float3 arithmetic that their CUDA build gets from tiny-cuda-nn / helper
// headers (absent from /root/reference); these few operators stand in for those.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include "cuda_shim.h"
inline float __saturatef(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
inline float3 operator+(const float3& a, const float3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(const float3& a, const float3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator-(const float3& a) { return {-a.x, -a.y, -a.z}; }
inline float3 operator*(const float3& a, const float3& b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator/(const float3& a, const float3& b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline float3 make_float3(float s) { return {s, s, s}; }
inline float4 make_float4(float s) { return {s, s, s, s}; }
inline float3 make_float3(int s) { return {(float)s, (float)s, (float)s}; }
