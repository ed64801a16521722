// ref_gut_render.cpp — runs the reference's OWN 3DGUT kernels on the host, thread block by thread block:
//   projectOnTiles / render / renderBackward   threedgut_tracer/include/3dgut/kernels/cuda/renderers/gutRenderer.cuh:21-118, 225-273
//   GUTKBufferRenderer::eval / evalKBuffer / evalBackwardNoKBuffer / processHitParticle and the hit k-buffer
//                                              renderers/gutKBufferRenderer.cuh:25-353, 531-718
//   initializeRay / finalizeRay / initializeBackwardRay      kernels/cuda/common/rayPayload.cuh, rayPayloadBackward.cuh
//   BoundingBox::ray_intersect                               utils/bounding_box.h:89-130
//   ShRadiativeGaussianVolumetricFeaturesParticles           kernels/cuda/models/shRadiativeGaussianParticles.cuh (the REAL
//       accessor class: fetch, rotation(), featuresCustomFromBuffer -> radianceFromSpH, processHitBwd -> threedgut::processHitBwd,
//       the 32-lane gradient reductions and atomics)
//   GUTProjector::eval                                       renderers/gutProjector.cuh:217-322 (as in ref_projector.cpp, but
//       with the real particle class, so the per-particle radiance it precomputes is the reference's)
// all of it included through the reference's own configuration header 3dgut/threedgut.cuh with the -D set of
// setup_3dgut.py:64-95 under configs/render/3dgut.yaml, and executed by the fiber emulation of a CUDA block in
// ref_block_emul.inl (real barriers, votes and shuffles between 256 cooperating threads).
// What is NOT the reference's: the four forward entry points of the Slang-generated header (shim/threedgutSlang.cuh, a
// restatement — cross-checked against the reference's CUDA twin by ref_gut_standin_max_error below), the tiny-cuda-nn types
// (shim/tiny-cuda-nn), and the host driver code in this file, which follows the launch sites of src/gutRenderer.cu:258-455.
// The K = 0 backward (the training path) reaches no Slang code at all.  TEST INFRASTRUCTURE ONLY.
#include <math.h>
#include <algorithm>
#include <vector>
#include "shim/cuda_shim.h"
#include "shim/gut_shim.h"
#include <tiny-cuda-nn/common.h>
#include "ref_block_emul.inl"

// ---- the -D set of setup_3dgut.py:47-95 for configs/render/3dgut.yaml (degree and K come from the Makefile) ----------------
#ifndef FEATURE_TRANSFORM_TYPE   // (the NHT build passes its feature macros from the Makefile: setup_3dgut.py:47-57 for model.feature_type nht)
#define PARTICLE_FEATURE_DIM 48
#define RAY_FEATURE_DIM 3
#define FEATURE_TRANSFORM_TYPE 0
#endif
#define PARTICLE_FEATURE_HALF 0
#define FEATURE_OUTPUT_HALF 0
#define PARTICLE_RADIANCE_NUM_COEFFS 16
#define GAUSSIAN_PARTICLE_MIN_KERNEL_DENSITY 0.0113f
#define GAUSSIAN_PARTICLE_MIN_ALPHA (1.0f / 255.0f)
#define GAUSSIAN_PARTICLE_MAX_ALPHA 0.99f
#define GAUSSIAN_PARTICLE_ENABLE_NORMAL false
#define GAUSSIAN_PARTICLE_SURFEL false
#define GAUSSIAN_MIN_TRANSMITTANCE_THRESHOLD 0.0001f
#define GAUSSIAN_ENABLE_HIT_COUNT true
#define GAUSSIAN_N_ROLLING_SHUTTER_ITERATIONS 5
#define GAUSSIAN_GLOBAL_Z_ORDER true
#ifndef FINE_GRAINED_LOAD_BALANCING   // true in the K = 0 builds: adds renderBalanced next to render (Makefile)
#define FINE_GRAINED_LOAD_BALANCING false
#endif
#define GAUSSIAN_UT_ALPHA 1.0f
#define GAUSSIAN_UT_BETA 2.0f
#define GAUSSIAN_UT_KAPPA 0.0f
#define GAUSSIAN_UT_DELTA 1.7320508075688772f
#define GAUSSIAN_UT_IN_IMAGE_MARGIN_FACTOR 0.1f
#define GAUSSIAN_UT_REQUIRE_ALL_SIGMA_POINTS_VALID false
#define GAUSSIAN_RECT_BOUNDING true
#define GAUSSIAN_TIGHT_OPACITY_BOUNDING true
#define GAUSSIAN_TILE_BASED_CULLING true
#define REF_REAL_BOUNDING_BOX 1

static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline unsigned min(unsigned a, int b) { return b < 0 ? 0u : (a < (unsigned)b ? a : (unsigned)b); }
using tcnn::length;
using tcnn::min;
using tcnn::sqrt;

#include <3dgut/threedgut.cuh>

using namespace threedgut;

namespace {
TSensorState sensor_state(const float* pose_start7, const float* pose_end7) {
    TSensorState st;
    st.startTimestamp = 0; st.endTimestamp = 1;
    for (int i = 0; i < 7; ++i) { st.startPose[i] = pose_start7[i]; st.endPose[i] = pose_end7[i]; }
    return st;
}
// camera parameter block as in ref_camera.cpp / ref_projector.cpp: prm[0..32]
TSensorModel sensor_model(int model, int shutter, const float* prm) {
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
    return m;
}
struct GlobalValues { int32_t pad; int32_t sh_degree; };   // FeatureShDegreeValueOffset = 4 bytes (threedgut.cuh:29)
template <class F>
void launch(unsigned gx, unsigned gy, unsigned bx, unsigned by, F&& kernel) {
    gridDim = {gx, gy, 1}; blockDim = {bx, by, 1};
    for (unsigned y = 0; y < gy; ++y)
        for (unsigned x = 0; x < gx; ++x) {
            blockIdx = {x, y, 1};
            emu::run_block(kernel);
        }
}
}  // namespace

extern "C" {

int ref_gut_k_buffer_size(void) { return GAUSSIAN_K_BUFFER_SIZE; }
int ref_gut_kernel_degree(void) { return GAUSSIAN_PARTICLE_KERNEL_DEGREE; }

// projectOnTiles with the launch geometry of gutRenderer.cu:258-288; any camera model / shutter type (prm as in ref_camera.cpp)
void ref_gut_project(int model, int shutter, int width, int height, const float* prm, const float* pose_start7, const float* pose_end7, uint32_t n,
                     const float* density12, const float* sph, int sh_degree, uint32_t* tiles_count, float* proj_pos, float* conic_opacity,
                     float* extent, float* depth, float* features, int* visibility) {
    const TSensorModel m = sensor_model(model, shutter, prm);
    const TSensorState st = sensor_state(pose_start7, pose_end7);
    const tcnn::uvec2 tileGrid((uint32_t)(width + 15) / 16, (uint32_t)(height + 15) / 16);
    const TSensorPose sensorPose = interpolatedSensorPose(st.startPose, st.endPose, 0.5f);
    const tcnn::vec3 sensorWorldPosition = sensorPoseInverse(sensorPose).slice<0, 3>();
    const tcnn::mat4x3 viewMatrix = sensorPoseToMat(sensorPose);
    GlobalValues gv = {0, sh_degree};
    const uint64_t handles[3] = {(uint64_t)&gv, (uint64_t)density12, (uint64_t)sph};
    launch((n + 255) / 256, 1, 256, 1, [&] {
        projectOnTiles(tileGrid, n, tcnn::ivec2(width, height), m, sensorWorldPosition, viewMatrix, st, tiles_count,
                       reinterpret_cast<tcnn::vec2*>(proj_pos), reinterpret_cast<tcnn::vec4*>(conic_opacity),
                       reinterpret_cast<tcnn::vec2*>(extent), depth, features, visibility, handles);
    });
}

// render (gutRenderer.cu:398-413): tile_ranges [tiles,2], sorted_idx [I], features [n,3] = the projection's precomputed radiance
void ref_gut_render_fwd(int width, int height, const float* pose_start7, const float* pose_end7, const float* aabb_min3, const float* aabb_max3,
                        uint32_t n, const float* density12, const float* sph, int sh_degree, const uint32_t* tile_ranges,
                        const uint32_t* sorted_idx, const float* features, const float* ray_o, const float* ray_d, float* out_feat_density,
                        float* out_hit_distance, float* out_hit_count) {
    RenderParameters params;
    params.id = 0;
    params.resolution = tcnn::ivec2(width, height);
    params.hitTransmittance = 0.f;
    params.objectAABB.min = tcnn::vec3(aabb_min3[0], aabb_min3[1], aabb_min3[2]);
    params.objectAABB.max = tcnn::vec3(aabb_max3[0], aabb_max3[1], aabb_max3[2]);
    params.sensorState = sensor_state(pose_start7, pose_end7);
    const TSensorPose sensorPose = interpolatedSensorPose(params.sensorState.startPose, params.sensorState.endPose, 0.5f);
    const TSensorPose sensorPoseInv = sensorPoseInverse(sensorPose);
    GlobalValues gv = {0, sh_degree};
    const uint64_t handles[3] = {(uint64_t)&gv, (uint64_t)density12, (uint64_t)sph};
    launch((width + 15) / 16, (height + 15) / 16, 16, 16, [&] {
        render(params, reinterpret_cast<const tcnn::uvec2*>(tile_ranges), sorted_idx, reinterpret_cast<const tcnn::vec3*>(ray_o),
               reinterpret_cast<const tcnn::vec3*>(ray_d), sensorPoseToMat(sensorPoseInv), out_hit_count, out_hit_distance, out_feat_density,
               nullptr, nullptr, nullptr, features, handles);
    });
}

#if FINE_GRAINED_LOAD_BALANCING
// renderBalanced (gutRenderer.cu:380-397): the reference's other forward, `render.splat.fine_grained_load_balancing: true` —
// one 32-lane warp per pixel, 2x2-pixel virtual tiles, warp-level prefix products instead of the sequential loop
void ref_gut_render_fwd_balanced(int width, int height, const float* pose_start7, const float* pose_end7, const float* aabb_min3,
                                 const float* aabb_max3, uint32_t n, const float* density12, const float* sph, int sh_degree,
                                 const uint32_t* tile_ranges, const uint32_t* sorted_idx, const float* features, const float* ray_o,
                                 const float* ray_d, float* out_feat_density, float* out_hit_distance, float* out_hit_count) {
    RenderParameters params;
    params.id = 0;
    params.resolution = tcnn::ivec2(width, height);
    params.hitTransmittance = 0.f;
    params.objectAABB.min = tcnn::vec3(aabb_min3[0], aabb_min3[1], aabb_min3[2]);
    params.objectAABB.max = tcnn::vec3(aabb_max3[0], aabb_max3[1], aabb_max3[2]);
    params.sensorState = sensor_state(pose_start7, pose_end7);
    const TSensorPose sensorPose = interpolatedSensorPose(params.sensorState.startPose, params.sensorState.endPose, 0.5f);
    const TSensorPose sensorPoseInv = sensorPoseInverse(sensorPose);
    GlobalValues gv = {0, sh_degree};
    const uint64_t handles[3] = {(uint64_t)&gv, (uint64_t)density12, (uint64_t)sph};
    const tcnn::uvec2 tileGrid((uint32_t)(width + 15) / 16, (uint32_t)(height + 15) / 16);
    launch(tileGrid.x * tileGrid.y * GUTParameters::Tiling::VirtualTilesPerTile, 1, GUTParameters::Tiling::FineGrainedThreadsPerBlock, 1, [&] {
        renderBalanced(params, reinterpret_cast<const tcnn::uvec2*>(tile_ranges), sorted_idx, reinterpret_cast<const tcnn::vec3*>(ray_o),
                       reinterpret_cast<const tcnn::vec3*>(ray_d), sensorPoseToMat(sensorPoseInv), out_hit_count, out_hit_distance,
                       out_feat_density, nullptr, nullptr, nullptr, features, handles, tileGrid);
    });
}
#endif

// renderBackward (gutRenderer.cu:472-505): g_density12 [n,12] and g_features [n,3] must arrive zeroed (gutRenderer.cu:458-466)
void ref_gut_render_bwd(int width, int height, const float* pose_start7, const float* pose_end7, const float* aabb_min3, const float* aabb_max3,
                        uint32_t n, const float* density12, const float* sph, int sh_degree, const uint32_t* tile_ranges,
                        const uint32_t* sorted_idx, const float* features, const float* ray_o, const float* ray_d, const float* feat_density,
                        const float* grad_feat_density, const float* hit_distance, const float* grad_hit_distance, float* g_density12,
                        float* g_sph, float* g_features) {
    RenderParameters params;
    params.id = 0;
    params.resolution = tcnn::ivec2(width, height);
    params.hitTransmittance = 0.f;
    params.objectAABB.min = tcnn::vec3(aabb_min3[0], aabb_min3[1], aabb_min3[2]);
    params.objectAABB.max = tcnn::vec3(aabb_max3[0], aabb_max3[1], aabb_max3[2]);
    params.sensorState = sensor_state(pose_start7, pose_end7);
    const TSensorPose sensorPose = interpolatedSensorPose(params.sensorState.startPose, params.sensorState.endPose, 0.5f);
    const TSensorPose sensorPoseInv = sensorPoseInverse(sensorPose);
    GlobalValues gv = {0, sh_degree};
    const uint64_t handles[3] = {(uint64_t)&gv, (uint64_t)density12, (uint64_t)sph};
    const uint64_t grad_handles[2] = {(uint64_t)g_density12, (uint64_t)g_sph};
    launch((width + 15) / 16, (height + 15) / 16, 16, 16, [&] {
        renderBackward(params, reinterpret_cast<const tcnn::uvec2*>(tile_ranges), sorted_idx, reinterpret_cast<const tcnn::vec3*>(ray_o),
                       reinterpret_cast<const tcnn::vec3*>(ray_d), sensorPoseToMat(sensorPoseInv), hit_distance, grad_hit_distance, feat_density,
                       grad_feat_density, nullptr, nullptr, nullptr, nullptr, nullptr, features, handles, nullptr, nullptr, nullptr,
                       g_features, grad_handles);
    });
}

// Cross-check of the Slang stand-in (shim/threedgutSlang.cuh) against the reference's hand-written CUDA twin of the same math,
// threedgut::processHitFwd<DEG, false, false> (gaussianParticles.cuh:350-421): for each of n (ray, particle) pairs, one hit
// integrated into state {T, rgb, depth} both ways; returns the largest absolute difference over all state components and
// writes how many pairs were accepted by each side.
int ref_gut_ray_feature_dim(void) { return RAY_FEATURE_DIM; }
int ref_gut_particle_feature_dim(void) { return PARTICLE_FEATURE_DIM; }

#if FEATURE_TRANSFORM_TYPE == 0
float ref_gut_standin_max_error(uint32_t n, const float* ray_o, const float* ray_d, const float* density12, const float* feat3,
                                uint32_t* accepted_standin, uint32_t* accepted_twin) {
    float worst = 0.f;
    *accepted_standin = *accepted_twin = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const float3 o = make_float3(ray_o[3 * i], ray_o[3 * i + 1], ray_o[3 * i + 2]);
        const float3 d = make_float3(ray_d[3 * i], ray_d[3 * i + 1], ray_d[3 * i + 2]);
        ParticleDensity p;
        std::memcpy(&p, density12 + 12 * (size_t)i, sizeof(p));
        // the renderer hands both sides the clamped per-particle radiance (gutKBufferRenderer.cuh:215, :661)
        const float fc[3] = {std::max(feat3[3 * i], 0.f), std::max(feat3[3 * i + 1], 0.f), std::max(feat3[3 * i + 2], 0.f)};
        // the twin
        float T1 = 0.7f, depth1 = 0.25f;
        float3 rad1 = make_float3(0.1f, 0.2f, 0.3f);
        const bool acc1 = processHitFwd<GAUSSIAN_PARTICLE_KERNEL_DEGREE, false, false>(
            o, d, 0, &p, fc, GAUSSIAN_PARTICLE_MIN_KERNEL_DENSITY, GAUSSIAN_PARTICLE_MIN_ALPHA, 0, &T1, &rad1, &depth1, nullptr);
        // the stand-in, called the way evalKBuffer / processHitParticle call it
        float T2 = 0.7f, depth2 = 0.25f;
        FixedArray<float, 3> rad2 = {{0.1f, 0.2f, 0.3f}};
        gaussianParticle_RawParameters_0* rows = reinterpret_cast<gaussianParticle_RawParameters_0*>(&p);
        const gaussianParticle_Parameters_0 prm = particleDensityParameters(0, {rows, nullptr, false});
        float alpha = 0.f, hitT = 0.f;
        float3 canonical = make_float3(0.f, 0.f, 0.f);
        const bool acc2 = particleDensityHit(o, d, prm, &alpha, &hitT, &canonical, false, nullptr);
        if (acc2) {
            const float w = particleDensityIntegrateHit(alpha, &T2, hitT, &depth2, false, make_float3(0, 0, 0), nullptr);
            FixedArray<float, 3> f = {{fc[0], fc[1], fc[2]}};
            particleFeaturesIntegrateFwd(w, f, &rad2);
        }
        *accepted_twin += acc1; *accepted_standin += acc2;
        if (acc1 != acc2) { worst = std::max(worst, 1.f); continue; }
        worst = std::max({worst, std::fabs(T1 - T2), std::fabs(depth1 - depth2), std::fabs(rad1.x - rad2[0]), std::fabs(rad1.y - rad2[1]),
                          std::fabs(rad1.z - rad2[2])});
    }
    return worst;
}
#endif

}  // extern "C"
