// ref_projector.cpp — runs the reference's OWN projection stage on the host, one particle at a time:
//   GUTProjector::eval (threedgut_tracer/include/3dgut/kernels/cuda/renderers/gutProjector.cuh:217-322) with everything it
//   calls there — unscentedParticleProjection (:124-215), computeProjectedExtentConicOpacity (:81-116),
//   computeTileSpaceBBox (:32-43), tileMinParticlePowerResponse (:49-78) — and, below it, cameraProjections.cuh /
//   sensors.h as in ref_camera.cpp.  Compiled where the header lies with clang++ -fdelayed-template-parsing (member
//   functions of the class template that are never instantiated, e.g. evalBackward, are then never parsed: they depend on
//   Slang-generated code that is not part of the checkout).
// What is NOT the reference's here, and therefore not pinned by this library: the `Particles` accessor class (the reference's
// one, shRadiativeGaussianParticles.cuh, is built on Slang output) — ShimParticles restates position / scale / opacity
// (plain fields of ParticleDensity) and rotation() (columns = the ellipsoid's world axes, shRadiativeGaussianParticles.cuh:
// 88-91 / SURVEY.md A1) — and the parameter structs, which carry the values of threedgut.cuh:54-81 under the default
// configs/render/3dgut.yaml.  TEST INFRASTRUCTURE ONLY (tests/golden/projector.npz, tests/test_oracle_cpu.py).
#include <math.h>
#include <algorithm>
#include <vector>
#include "shim/cuda_shim.h"
#include <tiny-cuda-nn/common.h>

// CUDA built-ins the projection code uses
struct ShimDim3 { unsigned x, y, z; };
static thread_local ShimDim3 blockIdx = {0, 0, 0}, blockDim = {1, 1, 1}, threadIdx = {0, 0, 0};
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __saturatef(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }
static inline unsigned min(unsigned a, int b) { return b < 0 ? 0u : (a < (unsigned)b ? a : (unsigned)b); }
using tcnn::length;
using tcnn::min;
using tcnn::sqrt;

// values of threedgut.cuh:54-81 with the defaults of configs/render/3dgut.yaml (setup_3dgut.py:41-95)
struct TGUTProjectorParams {
    static constexpr float ParticleMinSensorZ = 0.2;
    static constexpr float CovarianceDilation = 0.3;
    static constexpr float AlphaThreshold = 1.0f / 255.0f;
    static constexpr bool TightOpacityBounding = true;
    static constexpr bool RectBounding = true;
    static constexpr bool TileCulling = true;
    static constexpr bool PerRayParticleFeatures = false;
    static constexpr float MaxDepthValue = 3.4028235e+38;
    static constexpr bool GlobalZOrder = true;
    static constexpr bool BackwardProjection = false;
    static constexpr bool MipSplattingScaling = true;
};
struct TGUTProjectionParams {
    static constexpr int NRollingShutterIterations = 5;
    static constexpr int D = 3;
    static constexpr float Alpha = 1.0f;
    static constexpr float Beta = 2.0f;
    static constexpr float Kappa = 0.0f;
    static constexpr float Delta = 1.7320508075688772f;  // sqrt(Alpha^2 (D + Kappa))
    static constexpr float ImageMarginFactor = 0.1f;
    static constexpr bool RequireAllSigmaPoints = false;
};

#include <3dgut/kernels/cuda/renderers/gutProjector.cuh>

using namespace threedgut;

struct ShimParticles {
    using TFeaturesVec = tcnn::vec3;
    struct DensityParameters {
        tcnn::vec3 position;
        float density;
        tcnn::vec4 quaternion;  // (w, x, y, z)
        tcnn::vec3 scale;
        float pad;
    };
    const DensityParameters* rows = nullptr;
    void initializeDensity(MemoryHandles h) { rows = h.bufferPtr<const DensityParameters>(0); }
    DensityParameters fetchDensityParameters(uint32_t i) const { return rows[i]; }
    float opacity(const DensityParameters& p) const { return p.density; }
    const tcnn::vec3& position(const DensityParameters& p) const { return p.position; }
    const tcnn::vec3& scale(const DensityParameters& p) const { return p.scale; }
    tcnn::mat3 rotation(const DensityParameters& p) const { return tcnn::to_mat3(tcnn::quat{p.quaternion.x, p.quaternion.y, p.quaternion.z, p.quaternion.w}); }
    void initializeFeatures(MemoryHandles) {}
    template <bool B>
    TFeaturesVec featuresCustomFromBuffer(uint32_t, const tcnn::vec3&) const { return tcnn::vec3::zero(); }  // SH is pinned by per_hit_deg*.npz
};
using Projector = GUTProjector<ShimParticles, TGUTProjectorParams, TGUTProjectionParams>;

extern "C" {

// camera parameters as in ref_camera.cpp; density12 = [N,12] ParticleDensity rows
void ref_project_particles(int model, int shutter, int width, int height, const float* prm, const float* pose_start7, const float* pose_end7,
                           uint32_t n, const float* density12, uint32_t* tiles_count, float* proj_pos, float* conic_opacity, float* extent,
                           float* depth, int* visibility) {
    TSensorModel m;
    m.shutterType = (TSensorModel::ShutterType)shutter;
    if (model == 0) {
        m.modelType = TSensorModel::OpenCVPinholeModel;
        auto& q = m.ocvPinholeParams;
        q.principalPoint = tcnn::vec2(prm[0], prm[1]); q.focalLength = tcnn::vec2(prm[2], prm[3]);
        for (int i = 0; i < 6; ++i) q.radialCoeffs[i] = prm[4 + i];
        q.tangentialCoeffs = tcnn::vec2(prm[10], prm[11]);
        q.thinPrismCoeffs = tcnn::vec4(prm[12], prm[13], prm[14], prm[15]);
    } else if (model == 1) {
        m.modelType = TSensorModel::OpenCVFisheyeModel;
        auto& q = m.ocvFisheyeParams;
        q.principalPoint = tcnn::vec2(prm[0], prm[1]); q.focalLength = tcnn::vec2(prm[2], prm[3]);
        q.radialCoeffs = tcnn::vec4(prm[4], prm[5], prm[6], prm[7]);
        q.maxAngle = prm[16];
    } else {
        m.modelType = TSensorModel::FThetaModel;
        auto& q = m.fthetaParams;
        q.principalPoint = tcnn::vec2(prm[0], prm[1]);
        q.referencePoly = prm[17] != 0.f ? FThetaProjectionParameters::ANGLE_TO_PIXELDIST : FThetaProjectionParameters::PIXELDIST_TO_ANGLE;
        for (int i = 0; i < 6; ++i) { q.pixeldistToAnglePoly[i] = prm[18 + i]; q.angleToPixeldistPoly[i] = prm[24 + i]; }
        q.maxAngle = prm[16];
        for (int i = 0; i < 3; ++i) q.linear_cde[i] = prm[30 + i];
    }
    TSensorState st;
    st.startTimestamp = 0; st.endTimestamp = 1;
    for (int i = 0; i < 7; ++i) { st.startPose[i] = pose_start7[i]; st.endPose[i] = pose_end7[i]; }
    // the call site of gutRenderer.cu:258-288
    const tcnn::uvec2 tileGrid((uint32_t)(width + 15) / 16, (uint32_t)(height + 15) / 16);
    const TSensorPose sensorPose = interpolatedSensorPose(st.startPose, st.endPose, 0.5f);
    const tcnn::vec3 sensorWorldPosition = sensorPoseInverse(sensorPose).slice<0, 3>();
    const tcnn::mat4x3 viewMatrix = sensorPoseToMat(sensorPose);
    const uint64_t handle = reinterpret_cast<uint64_t>(density12);
    MemoryHandles mh{&handle};
    blockDim.x = 1; threadIdx.x = 0;
    std::vector<float> features(3 * (size_t)n);   // precomputed radiance: ShimParticles returns zeros (SH is pinned elsewhere)
    for (uint32_t i = 0; i < n; ++i) {
        blockIdx.x = i;
        Projector::eval(tileGrid, n, tcnn::ivec2(width, height), m, sensorWorldPosition, viewMatrix, st, tiles_count,
                        reinterpret_cast<tcnn::vec2*>(proj_pos), reinterpret_cast<tcnn::vec4*>(conic_opacity),
                        reinterpret_cast<tcnn::vec2*>(extent), depth, features.data(), visibility, mh);
    }
}

// GUTProjector::expand (gutProjector.cuh:324-388) for every particle: the unsorted (tile << 32 | depth bits, particle) lists.
// offsets = inclusive prefix sum of tiles_count (gutRenderer.cu:302-310).
void ref_expand_particles(int width, int height, uint32_t n, const uint32_t* offsets, const float* proj_pos, const float* conic_opacity,
                          const float* extent, const float* depth, uint64_t* keys, uint32_t* idx) {
    const tcnn::uvec2 tileGrid((uint32_t)(width + 15) / 16, (uint32_t)(height + 15) / 16);
    TSensorModel m;
    TSensorState st;
    MemoryHandles mh{nullptr};
    blockDim.x = 1; threadIdx.x = 0;
    for (uint32_t i = 0; i < n; ++i) {
        blockIdx.x = i;
        Projector::expand(tileGrid, (int)n, tcnn::ivec2(width, height), m, st, offsets, reinterpret_cast<const tcnn::vec2*>(proj_pos),
                          reinterpret_cast<const tcnn::vec4*>(conic_opacity), reinterpret_cast<const tcnn::vec2*>(extent), depth, mh, keys, idx);
    }
}

}  // extern "C"
