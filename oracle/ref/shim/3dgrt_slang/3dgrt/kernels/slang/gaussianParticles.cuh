// gaussianParticles.cuh — host STAND-IN for the header slangc generates from threedgrt_tracer/include/3dgrt/kernels/slang/ when the
// reference's playground builds (threedgrut_playground/setup_playground.py:40-75).  The generated file is not part of the checkout
// and slangc is not in this image, so the entry points that 3dgrt/kernels/cuda/3dgrtTracer.cuh calls are RESTATED here from the
// Slang sources (cited per function), with the SIGNATURES 3dgrtTracer.cuh:165-195, 206-224 uses (that header was written against a
// generated file whose processHit export has no `canonicalIntersection` output and which exports a "generic" feature integration;
// the .slang files of the checkout differ in exactly those two places).  This file is therefore NOT reference code and pins nothing
// by itself: ref_playground.cpp cross-checks it against the reference's hand-written CUDA twin (processHit / intersectInstanceParticle,
// 3dgrt/kernels/cuda/gaussianParticles.cuh), and what the library built on it pins is everything AROUND these calls — the path
// loop, the closest-hit dispatch, the materials, the volumetric rounds.  The vector operators at the top are the ones the Slang CUDA
// prelude provides (playground/kernels/cuda/mathUtils.cuh leaves exactly these out).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cmath>
#include <cstdint>

// ---- what the Slang CUDA prelude supplies --------------------------------------------------------------------------------
inline float3 make_float3(float a) { return make_float3(a, a, a); }
inline float3 make_float3(float4 a) { return make_float3(a.x, a.y, a.z); }
inline float4 make_float4(float a) { return make_float4(a, a, a, a); }
inline float2 operator/(const float2& a, const float2& b) { return make_float2(a.x / b.x, a.y / b.y); }
inline float2 operator*(const float2& a, const float2& b) { return make_float2(a.x * b.x, a.y * b.y); }
inline float2 operator+(const float2& a, const float2& b) { return make_float2(a.x + b.x, a.y + b.y); }
inline float2 operator-(const float2& a, const float2& b) { return make_float2(a.x - b.x, a.y - b.y); }
inline float2 operator-(const float2& a) { return make_float2(-a.x, -a.y); }
inline float3 operator/(const float3& a, const float3& b) { return make_float3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline float3 operator*(const float3& a, const float3& b) { return make_float3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float3 operator+(const float3& a, const float3& b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float3 operator-(const float3& a, const float3& b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float3 operator-(const float3& a) { return make_float3(-a.x, -a.y, -a.z); }
inline float4 operator/(const float4& a, const float4& b) { return make_float4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
inline float4 operator*(const float4& a, const float4& b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline float4 operator+(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline float4 operator-(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline float4 operator/(const float4& a, float b) { return make_float4(a.x / b, a.y / b, a.z / b, a.w / b); }
inline float4 operator*(const float4& a, float b) { return make_float4(a.x * b, a.y * b, a.z * b, a.w * b); }
inline float4 operator+(const float4& a, float b) { return make_float4(a.x + b, a.y + b, a.z + b, a.w + b); }
inline float4 operator-(const float4& a, float b) { return make_float4(a.x - b, a.y - b, a.z - b, a.w - b); }
inline float4 operator*(float a, const float4& b) { return make_float4(a * b.x, a * b.y, a * b.z, a * b.w); }
inline float4 operator-(const float4& a) { return make_float4(-a.x, -a.y, -a.z, -a.w); }

// ---- gaussianParticles.slang:19-43 -------------------------------------------------------------------------------------------
struct gaussianParticle_RawParameters_0 {
    float3 position_0;
    float density_0;
    float4 quaternion_0;
    float3 scale_0;
    float padding_0;
};
struct gaussianParticle_RawParametersBuffer_0 {
    gaussianParticle_RawParameters_0* _dataPtr_0;
    gaussianParticle_RawParameters_0* _gradPtr_0;
    bool exclusiveGradient_0;
};
struct gaussianParticle_CommonParameters_0 {
    gaussianParticle_RawParametersBuffer_0 parametersBuffer_0;
};

namespace grt_slang_standin {
inline float dot3(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
struct Rot { float3 r0, r1, r2; };
inline float3 mul33(const Rot& m, const float3& v) { return {dot3(m.r0, v), dot3(m.r1, v), dot3(m.r2, v)}; }
// transforms.rotationMatrixTranspose (kernels/slang/common/transforms.slang:21-39; quaternion stored (r, x, y, z))
inline Rot rotation_transpose(const float4& q) {
    const float xx = q.y * q.y, yy = q.z * q.z, zz = q.w * q.w;
    const float xy = q.y * q.z, xz = q.y * q.w, yz = q.z * q.w;
    const float rx = q.x * q.y, ry = q.x * q.z, rz = q.x * q.w;
    Rot m;
    m.r0 = {1.f - 2.f * (yy + zz), 2.f * (xy + rz), 2.f * (xz - ry)};
    m.r1 = {2.f * (xy - rz), 1.f - 2.f * (xx + zz), 2.f * (yz + rx)};
    m.r2 = {2.f * (xz + ry), 2.f * (yz - rx), 1.f - 2.f * (xx + yy)};
    return m;
}
// canonicalRayMaxKernelResponse (gaussianParticles.slang:127-175), volumetric particles
inline float max_response(float grayDist) {
    switch (GAUSSIAN_PARTICLE_KERNEL_DEGREE) {
    case 8: { const float sq = grayDist * grayDist; return std::exp(-0.000685871056241f * sq * sq); }
    case 5: return std::exp(-0.0185185185185f * grayDist * grayDist * std::sqrt(grayDist));
    case 4: return std::exp(-0.0555555555556f * grayDist * grayDist);
    case 3: return std::exp(-0.166666666667f * grayDist * std::sqrt(grayDist));
    case 1: return std::exp(-1.5f * std::sqrt(grayDist));
    case 0: return std::max(1.f + -0.329630334487f * std::sqrt(grayDist), 0.f);
    default: return std::exp(-0.5f * grayDist);
    }
}
}  // namespace grt_slang_standin

// particleDensityProcessHitFwdFromBuffer (gaussianParticles.slang:404-425) -> processHitFromBuffer<false> (:284-316) -> hit (:208-242:
// cannonicalRay :101-116, canonicalRayMinSquaredDistance :118-132, canonicalRayMaxKernelResponse :134-175, canonicalRayIntersection
// :186-195) + integrateHit<false> (:244-273), front to back.  Normals are not enabled in the playground's default build.
inline float particleDensityProcessHitFwdFromBuffer(float3 rayOrigin, float3 rayDirection, uint32_t particleIdx, gaussianParticle_CommonParameters_0 common,
                                                    float* transmittance, float* integratedDepth, bool /*enableNormal*/, float3* /*integratedNormal*/) {
    using namespace grt_slang_standin;
    const gaussianParticle_RawParameters_0 raw = common.parametersBuffer_0._dataPtr_0[particleIdx];
    const Rot rotT = rotation_transpose(raw.quaternion_0);
    const float3 giscl = {1.0f / raw.scale_0.x, 1.0f / raw.scale_0.y, 1.0f / raw.scale_0.z};
    const float3 gposc = {rayOrigin.x - raw.position_0.x, rayOrigin.y - raw.position_0.y, rayOrigin.z - raw.position_0.z};
    const float3 gposcr = mul33(rotT, gposc);
    const float3 o = {giscl.x * gposcr.x, giscl.y * gposcr.y, giscl.z * gposcr.z};
    const float3 rayDirR = mul33(rotT, rayDirection);
    const float3 grdu = {giscl.x * rayDirR.x, giscl.y * rayDirR.y, giscl.z * rayDirR.z};
    const float inv_len = 1.0f / std::sqrt(dot3(grdu, grdu));
    const float3 d = {grdu.x * inv_len, grdu.y * inv_len, grdu.z * inv_len};
    const float3 gcrod = {d.y * o.z - d.z * o.y, d.z * o.x - d.x * o.z, d.x * o.y - d.y * o.x};
    const float maxResponse = max_response(dot3(gcrod, gcrod));
    const float alpha = std::min((float)GAUSSIAN_PARTICLE_MAX_ALPHA, maxResponse * raw.density_0);
    if (!((maxResponse > (float)GAUSSIAN_PARTICLE_MIN_KERNEL_DENSITY) && (alpha > (float)GAUSSIAN_PARTICLE_MIN_ALPHA))) return 0.0f;
    const float along = dot3(d, {-1.f * o.x, -1.f * o.y, -1.f * o.z});
    const float3 grds = {raw.scale_0.x * (d.x * along), raw.scale_0.y * (d.y * along), raw.scale_0.z * (d.z * along)};
    const float depth = std::sqrt(dot3(grds, grds));
    const float weight = alpha * *transmittance;
    *integratedDepth += depth * weight;
    *transmittance *= (1 - alpha);
    return weight;
}

#if defined(GRT_SLANG_RAYGEN_BUILD)
// ---- the Slang forward pipeline (referenceSlangOptix.cu:147-175), neural harmonic features or SH radiance ---------------------------
template <typename T, int N>
struct FixedArray {
    T m_data[N];
    T& operator[](int i) { return m_data[i]; }
    const T& operator[](int i) const { return m_data[i]; }
};
// particleDensityProcessHitFwdFromBuffer with the canonical intersection (gaussianParticles.slang:404-425 -> processHitFromBuffer<false>
// :284-316): the function above + canonicalIntersection = gro + grd (grd . -gro) (:181-190), written when the hit is accepted
inline float particleDensityProcessHitFwdFromBuffer(float3 rayOrigin, float3 rayDirection, uint32_t particleIdx, gaussianParticle_CommonParameters_0 common,
                                                    float* transmittance, float* integratedDepth, float3* canonicalIntersection, bool /*enableNormal*/,
                                                    float3* /*integratedNormal*/) {
    using namespace grt_slang_standin;
    const gaussianParticle_RawParameters_0 raw = common.parametersBuffer_0._dataPtr_0[particleIdx];
    const Rot rotT = rotation_transpose(raw.quaternion_0);
    const float3 giscl = {1.0f / raw.scale_0.x, 1.0f / raw.scale_0.y, 1.0f / raw.scale_0.z};
    const float3 gposc = {rayOrigin.x - raw.position_0.x, rayOrigin.y - raw.position_0.y, rayOrigin.z - raw.position_0.z};
    const float3 gposcr = mul33(rotT, gposc);
    const float3 o = {giscl.x * gposcr.x, giscl.y * gposcr.y, giscl.z * gposcr.z};
    const float3 rayDirR = mul33(rotT, rayDirection);
    const float3 grdu = {giscl.x * rayDirR.x, giscl.y * rayDirR.y, giscl.z * rayDirR.z};
    const float inv_len = 1.0f / std::sqrt(dot3(grdu, grdu));
    const float3 d = {grdu.x * inv_len, grdu.y * inv_len, grdu.z * inv_len};
    const float3 gcrod = {d.y * o.z - d.z * o.y, d.z * o.x - d.x * o.z, d.x * o.y - d.y * o.x};
    const float maxResponse = max_response(dot3(gcrod, gcrod));
    const float alpha = std::min((float)GAUSSIAN_PARTICLE_MAX_ALPHA, maxResponse * raw.density_0);
    if (!((maxResponse > (float)GAUSSIAN_PARTICLE_MIN_KERNEL_DENSITY) && (alpha > (float)GAUSSIAN_PARTICLE_MIN_ALPHA))) return 0.0f;
    const float along = dot3(d, {-1.f * o.x, -1.f * o.y, -1.f * o.z});
    const float3 cg = {d.x * along, d.y * along, d.z * along};
    *canonicalIntersection = {o.x + cg.x, o.y + cg.y, o.z + cg.z};
    const float3 grds = {raw.scale_0.x * cg.x, raw.scale_0.y * cg.y, raw.scale_0.z * cg.z};
    const float depth = std::sqrt(dot3(grds, grds));
    const float weight = alpha * *transmittance;
    *integratedDepth += depth * weight;
    *transmittance *= (1 - alpha);
    return weight;
}
#endif
#if defined(GRT_SLANG_RAYGEN_BUILD) && FEATURE_TRANSFORM_TYPE == 1
// particleFeaturesIntegrateFwdFromBuffer of the neural-harmonic-features model (neuralHarmonicFeaturesParticle.slang:253-270 ->
// integrateFeaturesFromBuffer<false> :213-228 -> featuresFromParametersBuffer :146-196).  No CUDA twin in the checkout: a restatement.
template <typename TElem>
inline void particleFeaturesIntegrateFwdFromBuffer(float3 /*incidentDirection*/, float3 canonicalPosition, float weight, uint32_t particleIdx,
                                                   TElem* featuresBufferPtr, int /*auxParam*/, FixedArray<float, RAY_FEATURE_DIM>* integrated) {
    if (!(weight > 0.0f)) return;
    constexpr int IPD = INTERP_POINT_FEATURE_DIM;
    const TElem* row = featuresBufferPtr + (size_t)particleIdx * PARTICLE_FEATURE_DIM;
    float base[IPD];
    for (int n = 0; n < IPD; ++n) base[n] = (float)row[n];
#if FEATURE_INTERPOLATION_SUPPORT == 1 && FEATURE_INTERPOLATION_TYPE == 0
    {
        using namespace grt_slang_standin;
        auto sub = [](float3 a, float3 b) { return float3{a.x - b.x, a.y - b.y, a.z - b.z}; };
        auto crs = [](float3 a, float3 b) { return float3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
        const float edge = 4.898979485566356f, faceHeight = 4.242640687119285f, faceInRadius = 1.4142135623730951f;
        const float3 v0 = {0.5f * edge, -faceInRadius, -1.0f}, v1 = {-0.5f * edge, -faceInRadius, -1.0f}, v2 = {0.0f, faceHeight - faceInRadius, -1.0f},
                     v3 = {0.0f, 0.0f, 3.0f};
        const float3 e1 = sub(v1, v0), e2 = sub(v2, v0), e3 = sub(v3, v0);
        const float3 c23 = crs(e2, e3);
        const float invDet = 1.0f / dot3(e1, c23);
        const float3 d = sub(canonicalPosition, v0);
        float w[4];
        w[1] = dot3(d, c23) * invDet; w[2] = dot3(e1, crs(d, e3)) * invDet; w[3] = dot3(e1, crs(e2, d)) * invDet;
        w[0] = 1.0f - w[1] - w[2] - w[3];
        for (int n = 0; n < IPD; ++n) base[n] *= w[0];
        for (int k = 1; k < 4; ++k)
            for (int n = 0; n < IPD; ++n) base[n] += w[k] * (float)row[k * IPD + n];
    }
#endif
#if FEATURE_ACTIVATION_TYPE == 2
    for (int k = 0; k < IPD; ++k)
        for (int f = 0; f < FEATURE_ACTIVATION_NUM_FREQUENCIES; ++f) {
            const float angle = base[k] * (float)(f + 1);
            (*integrated)[k * FEATURE_ACTIVATION_NUM_FREQUENCIES * 2 + f * 2 + 0] += std::sin(angle) * weight;
            (*integrated)[k * FEATURE_ACTIVATION_NUM_FREQUENCIES * 2 + f * 2 + 1] += std::cos(angle) * weight;
        }
#else
#error "ref_grt_trace_slang builds the sincos configuration"
#endif
}
#endif

// "particleFeaturesIntegrateFwdGeneric" (3dgrtTracer.cuh:185-191; not in the checkout's .slang files): for FEATURE_TRANSFORM_TYPE 0 it
// is integrateRadianceFromBuffer<false> (shRadiativeParticles.slang:117-130) -> sphericalHarmonics.decode (sphericalHarmonics.slang:21-64)
// with the RAY direction as the incident direction, front to back
inline void particleFeaturesIntegrateFwdGeneric(float3 dir, float weight, uint32_t particleIdx, const float* particleFeatures, unsigned sphDegree, float* integrated) {
    if (!(weight > 0.0f)) return;
    const float* c = particleFeatures + (size_t)particleIdx * 3 * PARTICLE_RADIANCE_NUM_COEFFS;
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                         -0.5900435899266435f};
    const int degree = (int)sphDegree;
    for (int ch = 0; ch < 3; ++ch) {
        auto K = [&](int k) { return c[3 * k + ch]; };
        float f = C0 * K(0);
        if (PARTICLE_RADIANCE_NUM_COEFFS >= 4 && degree > 0) {
            const float x = dir.x, y = dir.y, z = dir.z;
            f = f - C1 * y * K(1) + C1 * z * K(2) - C1 * x * K(3);
            if (PARTICLE_RADIANCE_NUM_COEFFS >= 9 && degree > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                f = f + C2[0] * xy * K(4) + C2[1] * yz * K(5) + C2[2] * (2.0f * zz - xx - yy) * K(6) + C2[3] * xz * K(7) + C2[4] * (xx - yy) * K(8);
                if (PARTICLE_RADIANCE_NUM_COEFFS >= 16 && degree > 2) {
                    f = f + C3[0] * y * (3.0f * xx - yy) * K(9) + C3[1] * xy * z * K(10) + C3[2] * y * (4.0f * zz - xx - yy) * K(11) +
                        C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * K(12) + C3[4] * x * (4.0f * zz - xx - yy) * K(13) + C3[5] * z * (xx - yy) * K(14) +
                        C3[6] * x * (xx - 3.0f * yy) * K(15);
                }
            }
        }
        integrated[ch] += std::max(f + 0.5f, 0.0f) * weight;
    }
}

#if defined(GRT_SLANG_RAYGEN_BUILD) && FEATURE_TRANSFORM_TYPE == 0
// particleFeaturesIntegrateFwdFromBuffer of the SH model (shRadiativeParticles.slang:117-130): the radiance decoded with the RAY
// direction as the incident direction, integrated front to back — the function above behind the Slang raygen's signature
template <typename TElem>
inline void particleFeaturesIntegrateFwdFromBuffer(float3 incidentDirection, float3 /*canonicalPosition*/, float weight, uint32_t particleIdx,
                                                   TElem* featuresBufferPtr, int sphDegree, FixedArray<float, RAY_FEATURE_DIM>* integrated) {
    static_assert(RAY_FEATURE_DIM == 3, "SH radiance integrates three channels");
    particleFeaturesIntegrateFwdGeneric(incidentDirection, weight, particleIdx, featuresBufferPtr, (unsigned)sphDegree, &(*integrated)[0]);
}
#endif

// particleDensityHitInstance (gaussianParticles.slang:525-541): the hit distance is the closest approach to the proxy's centre in its
// own (scaled) frame; canonicalRayMinSquaredDistance (:118-132, volumetric) on the NORMALISED direction
inline bool particleDensityHitInstance(float3 o, float3 dUn, float minHitDistance, float maxHitDistance, float maxParticleSquaredDistance, float* hitDistance) {
    using namespace grt_slang_standin;
    const float numerator = -dot3(o, dUn);
    const float denominator = 1.0f / dot3(dUn, dUn);
    *hitDistance = numerator * denominator;
    const float il = 1.0f / std::sqrt(dot3(dUn, dUn));
    const float3 d = {dUn.x * il, dUn.y * il, dUn.z * il};
    const float3 gcrod = {d.y * o.z - d.z * o.y, d.z * o.x - d.x * o.z, d.x * o.y - d.y * o.x};
    return (*hitDistance > minHitDistance) && (*hitDistance < maxHitDistance) && (dot3(gcrod, gcrod) < maxParticleSquaredDistance);
}
// particleDensityHitCustom (gaussianParticles.slang:489-523), the custom-primitive pipeline's test in world space: cannonicalRay (:103-117), the
// hit distance canonicalRayDistance (:180-186) = |scale * d (d . -o)| - UNSIGNED, unlike intersectCustomParticle's of the CUDA pipeline -,
// accepted within maxParticleSquaredDistance of the centre in the scale frame (canonicalRayMinSquaredDistance :118-132, volumetric)
inline bool particleDensityHitCustom(float3 rayOrigin, float3 rayDirection, int32_t particleIdx, gaussianParticle_CommonParameters_0 common, float minHitDistance,
                                     float maxHitDistance, float maxParticleSquaredDistance, float* hitDistance) {
    using namespace grt_slang_standin;
    const gaussianParticle_RawParameters_0 raw = common.parametersBuffer_0._dataPtr_0[particleIdx];
    const Rot rotT = rotation_transpose(raw.quaternion_0);
    const float3 giscl = {1.0f / raw.scale_0.x, 1.0f / raw.scale_0.y, 1.0f / raw.scale_0.z};
    const float3 gposc = {rayOrigin.x - raw.position_0.x, rayOrigin.y - raw.position_0.y, rayOrigin.z - raw.position_0.z};
    const float3 gposcr = mul33(rotT, gposc);
    const float3 o = {giscl.x * gposcr.x, giscl.y * gposcr.y, giscl.z * gposcr.z};
    const float3 rayDirR = mul33(rotT, rayDirection);
    const float3 grdu = {giscl.x * rayDirR.x, giscl.y * rayDirR.y, giscl.z * rayDirR.z};
    const float inv_len = 1.0f / std::sqrt(dot3(grdu, grdu));
    const float3 d = {grdu.x * inv_len, grdu.y * inv_len, grdu.z * inv_len};
    const float along = dot3(d, {-1.f * o.x, -1.f * o.y, -1.f * o.z});
    const float3 grds = {raw.scale_0.x * (d.x * along), raw.scale_0.y * (d.y * along), raw.scale_0.z * (d.z * along)};
    *hitDistance = std::sqrt(dot3(grds, grds));
    const float3 gcrod = {d.y * o.z - d.z * o.y, d.z * o.x - d.x * o.z, d.x * o.y - d.y * o.x};
    return (*hitDistance > minHitDistance) && (*hitDistance < maxHitDistance) && (dot3(gcrod, gcrod) < maxParticleSquaredDistance);
}
