// cutexture.h — host stand-in for threedgrut_playground/include/playground/cutexture.h (which wraps CUDA arrays and texture objects,
// a CUDA-runtime facility).  A texture object here is a pointer to a ShimTexture; tex2D restates the addressing the reference
// configures for every texture it creates (cutexture.h:54-60): normalised coordinates, clamp-to-edge, bilinear filtering, float
// elements.  CUDA's bilinear filter interpolates with 9-bit fixed-point weights (8 fractional bits, CUDA programming guide appendix
// "Linear Filtering"); this stand-in — like the C oracle and the HIP kernels — uses the float weights themselves, a difference of at
// most 1/512 of the texel-to-texel variation that no reference artefact pins.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cmath>
typedef unsigned long long cudaTextureObject_t;
struct ShimTexture {
    int height, width, channels;   // texel [y][x][c], row-major (the reference uploads torch tensors [H, W, C])
    const float* data;
};
inline void shim_tex_fetch(const ShimTexture* t, float u, float v, float out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0.f;
    if (!t || !t->data) return;
    // unnormalised texel coordinates, sample centres at +0.5; clamp-to-edge addressing
    const float x = u * (float)t->width - 0.5f, y = v * (float)t->height - 0.5f;
    const float fx = std::floor(x), fy = std::floor(y);
    const float ax = x - fx, ay = y - fy;
    auto cl = [](int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); };
    const int x0 = cl((int)fx, t->width), x1 = cl((int)fx + 1, t->width), y0 = cl((int)fy, t->height), y1 = cl((int)fy + 1, t->height);
    for (int c = 0; c < t->channels && c < 4; ++c) {
        const float t00 = t->data[((size_t)y0 * t->width + x0) * t->channels + c], t10 = t->data[((size_t)y0 * t->width + x1) * t->channels + c];
        const float t01 = t->data[((size_t)y1 * t->width + x0) * t->channels + c], t11 = t->data[((size_t)y1 * t->width + x1) * t->channels + c];
        out[c] = (1.f - ay) * ((1.f - ax) * t00 + ax * t10) + ay * ((1.f - ax) * t01 + ax * t11);
    }
}
template <class T> inline T tex2D(cudaTextureObject_t tex, float u, float v);
template <> inline float4 tex2D<float4>(cudaTextureObject_t tex, float u, float v) {
    float o[4];
    shim_tex_fetch(reinterpret_cast<const ShimTexture*>(tex), u, v, o);
    return make_float4(o[0], o[1], o[2], o[3]);
}
template <> inline float2 tex2D<float2>(cudaTextureObject_t tex, float u, float v) {
    float o[4];
    shim_tex_fetch(reinterpret_cast<const ShimTexture*>(tex), u, v, o);
    return make_float2(o[0], o[1]);
}
