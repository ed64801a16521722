// threedgutSlang.cuh — host STAND-IN for the header slangc generates from threedgut_tracer/include/3dgut/kernels/slang/
// when the reference builds (setup_3dgut.py:118-146).  The generated file is not part of the checkout and slangc is not in
// this image, so the handful of exported entry points that the host build of oracle/ref/ref_gut_render.cpp reaches are
// RESTATED here from the Slang sources (cited per function).  This file is therefore NOT reference code and pins nothing by
// itself: ref_gut_render.cpp cross-checks it against the reference's hand-written CUDA twin (threedgut::processHitFwd,
// kernels/cuda/models/gaussianParticles.cuh:350-421), and what the library built on it pins is everything AROUND these calls —
// the particle class, the tile loop, the k-buffer, ray set-up / write-out, and the whole K = 0 backward, which calls no Slang.
// Entry points the host build never reaches (Slang autodiff products) abort.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cstdio>
#include <cstdlib>

template <typename T, int N>
struct FixedArray {
    T m_data[N];
    T& operator[](int i) { return m_data[i]; }
    const T& operator[](int i) const { return m_data[i]; }
};

// gaussianParticles.slang:19-25 / :83-88 (a Slang float3x3 is three float3 rows)
struct gaussianParticle_RawParameters_0 {
    float3 position_0;
    float density_0;
    float4 quaternion_0;
    float3 scale_0;
    float padding_0;
};
struct SlangFloat3x3 { float3 rows[3]; };
struct gaussianParticle_Parameters_0 {
    float3 position_1;
    float3 scale_1;
    SlangFloat3x3 rotationT_0;
    float density_1;
};
// gaussianParticles.slang:27-35
struct gaussianParticle_RawParametersBuffer_0 {
    gaussianParticle_RawParameters_0* _dataPtr_0;
    gaussianParticle_RawParameters_0* _gradPtr_0;
    bool exclusiveGradient_0;
};
struct gaussianParticle_CommonParameters_0 {
    gaussianParticle_RawParametersBuffer_0 parametersBuffer_0;
};

namespace slang_standin {
inline float dot3(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 mul33(const SlangFloat3x3& m, const float3& v) { return {dot3(m.rows[0], v), dot3(m.rows[1], v), dot3(m.rows[2], v)}; }
[[noreturn]] inline void unreachable(const char* what) {
    std::fprintf(stderr, "threedgutSlang stand-in: %s is a Slang autodiff product that this host build does not provide\n", what);
    std::abort();
}
}  // namespace slang_standin

// particleDensityParameters (gaussianParticles.slang:344-352) -> fetchParameters (:90-100), rotationMatrixTranspose
// (kernels/slang/common/transforms.slang:21-39; quaternion stored (r, x, y, z), not normalised here)
inline gaussianParticle_Parameters_0 particleDensityParameters(uint32_t particleIdx, gaussianParticle_CommonParameters_0 common) {
    const gaussianParticle_RawParameters_0 raw = common.parametersBuffer_0._dataPtr_0[particleIdx];
    const float4 q = raw.quaternion_0;
    const float xx = q.y * q.y, yy = q.z * q.z, zz = q.w * q.w;
    const float xy = q.y * q.z, xz = q.y * q.w, yz = q.z * q.w;
    const float rx = q.x * q.y, ry = q.x * q.z, rz = q.x * q.w;
    gaussianParticle_Parameters_0 p;
    p.position_1 = raw.position_0;
    p.scale_1 = raw.scale_0;
    p.rotationT_0.rows[0] = {1.f - 2.f * (yy + zz), 2.f * (xy + rz), 2.f * (xz - ry)};
    p.rotationT_0.rows[1] = {2.f * (xy - rz), 1.f - 2.f * (xx + zz), 2.f * (yz + rx)};
    p.rotationT_0.rows[2] = {2.f * (xz + ry), 2.f * (yz - rx), 1.f - 2.f * (xx + yy)};
    p.density_1 = raw.density_0;
    return p;
}

// particleDensityHit (gaussianParticles.slang:354-373) -> hit (:208-242): cannonicalRay (:101-116), canonicalRayMinSquaredDistance
// (:112-125, volumetric branch), canonicalRayMaxKernelResponse (:127-175), canonicalRayIntersection (:186-195).  Normals are
// not enabled in this build (GAUSSIAN_PARTICLE_ENABLE_NORMAL false).
inline bool particleDensityHit(float3 rayOrigin, float3 rayDirection, const gaussianParticle_Parameters_0& prm, float* alpha, float* depth,
                               float3* canonicalIntersection, bool /*enableNormal*/, float3* /*normal*/) {
    using namespace slang_standin;
    const float3 giscl = {1.0f / prm.scale_1.x, 1.0f / prm.scale_1.y, 1.0f / prm.scale_1.z};
    const float3 gposc = {rayOrigin.x - prm.position_1.x, rayOrigin.y - prm.position_1.y, rayOrigin.z - prm.position_1.z};
    const float3 gposcr = mul33(prm.rotationT_0, gposc);
    const float3 o = {giscl.x * gposcr.x, giscl.y * gposcr.y, giscl.z * gposcr.z};
    const float3 rayDirR = mul33(prm.rotationT_0, rayDirection);
    const float3 grdu = {giscl.x * rayDirR.x, giscl.y * rayDirR.y, giscl.z * rayDirR.z};
    const float inv_len = 1.0f / std::sqrt(dot3(grdu, grdu));
    const float3 d = {grdu.x * inv_len, grdu.y * inv_len, grdu.z * inv_len};

    const float3 gcrod = {d.y * o.z - d.z * o.y, d.z * o.x - d.x * o.z, d.x * o.y - d.y * o.x};
    const float grayDist = dot3(gcrod, gcrod);
    float maxResponse;
    switch (GAUSSIAN_PARTICLE_KERNEL_DEGREE) {
    case 8: { const float sq = grayDist * grayDist; maxResponse = std::exp(-0.000685871056241f * sq * sq); break; }
    case 5: maxResponse = std::exp(-0.0185185185185f * grayDist * grayDist * std::sqrt(grayDist)); break;
    case 4: maxResponse = std::exp(-0.0555555555556f * grayDist * grayDist); break;
    case 3: maxResponse = std::exp(-0.166666666667f * grayDist * std::sqrt(grayDist)); break;
    case 1: maxResponse = std::exp(-1.5f * std::sqrt(grayDist)); break;
    case 0: maxResponse = std::max(1.f + -0.329630334487f * std::sqrt(grayDist), 0.f); break;
    default: maxResponse = std::exp(-0.5f * grayDist); break;
    }
    *alpha = std::min((float)GAUSSIAN_PARTICLE_MAX_ALPHA, maxResponse * prm.density_1);
    const bool acceptHit = (maxResponse > (float)GAUSSIAN_PARTICLE_MIN_KERNEL_DENSITY) && (*alpha > (float)GAUSSIAN_PARTICLE_MIN_ALPHA);
    if (acceptHit) {
        const float along = dot3(d, {-1.f * o.x, -1.f * o.y, -1.f * o.z});
        const float3 canonicalGrds = {d.x * along, d.y * along, d.z * along};
        *canonicalIntersection = {o.x + canonicalGrds.x, o.y + canonicalGrds.y, o.z + canonicalGrds.z};
        const float3 grds = {prm.scale_1.x * canonicalGrds.x, prm.scale_1.y * canonicalGrds.y, prm.scale_1.z * canonicalGrds.z};
        *depth = std::sqrt(dot3(grds, grds));
    }
    return acceptHit;
}

// particleDensityIntegrateHit (gaussianParticles.slang:375-392) -> integrateHit<false> (:244-273), front to back
inline float particleDensityIntegrateHit(float alpha, float* transmittance, float depth, float* integratedDepth, bool /*enableNormal*/,
                                         float3 /*normal*/, float3* /*integratedNormal*/) {
    const float weight = alpha * *transmittance;
    *integratedDepth += depth * weight;
    *transmittance *= (1 - alpha);
    return weight;
}

// particleFeaturesIntegrateFwd (shRadiativeParticles.slang:165-182) -> integrateRadiance<false> (:84-99)
template <int N>
inline void particleFeaturesIntegrateFwd(float weight, const FixedArray<float, N>& features, FixedArray<float, N>* integratedFeatures) {
    if (weight > 0.0f)
        for (int i = 0; i < N; ++i) (*integratedFeatures)[i] += features[i] * weight;
}

#if defined(FEATURE_TRANSFORM_TYPE) && FEATURE_TRANSFORM_TYPE == 1
// particleFeaturesFromBuffer of the NEURAL HARMONIC FEATURES model (kernels/slang/models/neuralHarmonicFeaturesParticle.slang:233-251 ->
// featuresFromParametersBuffer :146-196: fetchParametersFromBuffer :85-97, barycentricTetrahedronCanonical :117-127 over the canonical
// tetrahedron of :47-66, then the activation).  This model has NO hand-written CUDA twin in the checkout, so unlike the entry points
// above this restatement cannot be cross-checked against reference code: what libref_gut_render_nht pins is the renderer around it
// (tile loop, hit test, canonical intersection, integration order, ray set-up and write-out of RAY_FEATURE_DIM + 1 channels).
namespace slang_standin {
inline float3 sub3(const float3& a, const float3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 cross3(const float3& a, const float3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
}
template <typename TElem>
inline void particleFeaturesFromBuffer(uint32_t particleIdx, TElem* featuresBufferPtr, int /*auxParam*/, float3 /*incidentDirection*/,
                                       float3 canonicalPosition, FixedArray<float, RAY_FEATURE_DIM>* features) {
    using namespace slang_standin;
    constexpr int IPD = INTERP_POINT_FEATURE_DIM;
    const TElem* row = featuresBufferPtr + (size_t)particleIdx * PARTICLE_FEATURE_DIM;
    float base[IPD];
    for (int n = 0; n < IPD; ++n) base[n] = (float)row[n];
#if FEATURE_INTERPOLATION_SUPPORT == 1 && FEATURE_INTERPOLATION_TYPE == 0
    {
        const float edge = 4.898979485566356f, faceHeight = 4.242640687119285f, height = 4.0f, faceInRadius = 1.4142135623730951f, inRadius = 1.0f;
        const float3 v0 = {0.5f * edge, -faceInRadius, -1.0f}, v1 = {-0.5f * edge, -faceInRadius, -1.0f},
                     v2 = {0.0f, faceHeight - faceInRadius, -1.0f}, v3 = {0.0f, 0.0f, height - inRadius};
        const float3 e1 = sub3(v1, v0), e2 = sub3(v2, v0), e3 = sub3(v3, v0);
        const float3 c23 = cross3(e2, e3);
        const float invDet = 1.0f / dot3(e1, c23);
        const float3 d = sub3(canonicalPosition, v0);
        float w[4];
        w[1] = dot3(d, c23) * invDet;
        w[2] = dot3(e1, cross3(d, e3)) * invDet;
        w[3] = dot3(e1, cross3(e2, d)) * invDet;
        w[0] = 1.0f - w[1] - w[2] - w[3];
        for (int n = 0; n < IPD; ++n) base[n] *= w[0];
        for (int k = 1; k < 4; ++k)
            for (int n = 0; n < IPD; ++n) base[n] += w[k] * (float)row[k * IPD + n];
    }
#endif
#if FEATURE_ACTIVATION_TYPE == 0
    for (int i = 0; i < IPD; ++i) (*features)[i] = base[i];
#elif FEATURE_ACTIVATION_TYPE == 3
    for (int i = 0; i < IPD; ++i) (*features)[i] = std::max(0.0f, base[i]);
#elif FEATURE_ACTIVATION_TYPE == 2
    for (int k = 0; k < IPD; ++k)
        for (int f = 0; f < FEATURE_ACTIVATION_NUM_FREQUENCIES; ++f) {
            const float angle = base[k] * (float)(f + 1);
            (*features)[k * FEATURE_ACTIVATION_NUM_FREQUENCIES * 2 + f * 2 + 0] = std::sin(angle);
            (*features)[k * FEATURE_ACTIVATION_NUM_FREQUENCIES * 2 + f * 2 + 1] = std::cos(angle);
        }
#else   // siren: sin(b 2^f)
    for (int k = 0; k < IPD; ++k)
        for (int f = 0; f < FEATURE_ACTIVATION_NUM_FREQUENCIES; ++f)
            (*features)[k * FEATURE_ACTIVATION_NUM_FREQUENCIES + f] = std::sin(base[k] * std::ldexp(1.0f, f));
#endif
}
#endif

// ---- autodiff products: declared so that the reference's non-template kernels parse, never reached ------------------------
template <int N>
inline void particleFeaturesBwdToBuffer(uint32_t, float*, float*, int, bool, const FixedArray<float, N>&, float3, float3*) {
    slang_standin::unreachable("particleFeaturesBwdToBuffer");
}
inline void particleDensityIncidentDirectionBwdToBuffer(uint32_t, gaussianParticle_CommonParameters_0, float3, float3) {
    slang_standin::unreachable("particleDensityIncidentDirectionBwdToBuffer");
}
template <typename TElem, int N>
inline void particleFeaturesIntegrateBwdToBuffer(float3, float3, float3*, float, float*, uint32_t, TElem*, float*, int, bool, const FixedArray<float, N>&,
                                                 FixedArray<float, N>*, FixedArray<float, N>*) {
    slang_standin::unreachable("particleFeaturesIntegrateBwdToBuffer");
}
template <int N>
inline void particleFeaturesIntegrateBwd(float, float*, const FixedArray<float, N>&, FixedArray<float, N>*, FixedArray<float, N>*,
                                         FixedArray<float, N>*) {
    slang_standin::unreachable("particleFeaturesIntegrateBwd");
}
inline void particleDensityProcessHitBwdToBuffer(float3, float3, uint32_t, gaussianParticle_CommonParameters_0, float, float, float*, float*, float,
                                                 float*, float*, float3, bool, float3, float3*, float3*) {
    slang_standin::unreachable("particleDensityProcessHitBwdToBuffer");
}
