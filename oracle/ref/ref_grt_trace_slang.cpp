// ref_grt_trace_slang.cpp — the reference's 3DGRT forward programs of the SLANG pipeline (render.pipeline_type referenceSlang) in the
// neural-harmonic-features configuration, run on the host: __raygen__rg, trace(), __intersection__is and __anyhit__ah of
// threedgrt_tracer/src/kernels/cuda/referenceSlangOptix.cu, included as they lie, over the emulated OptiX of ref_grt_trace.cpp.
// What is NOT the reference's: OptiX (see ref_grt_trace.cpp) and the Slang-generated per-hit header, for which
// shim/3dgrt_slang/.../gaussianParticles.cuh restates the entry points the programs call (the density functions are cross-checked against
// the CUDA twin for the playground build; the feature model has no twin).  What this library pins is the program around them: the
// round loop, the k = 16 payload and its insertion chain, which hits are processed, the integration order and the write-out.
// -DREF_SLANG_SH builds the same programs with SH radiance (libref_grt_trace_slangsh_deg4.so, tests/golden/grt_trace_slang_sh.npz).
// TEST INFRASTRUCTURE ONLY: tests/golden/grt_trace_nht.npz, grt_trace_slang_sh.npz.
#include <math.h>
#include <type_traits>
#include <vector>
#include "shim/cuda_shim.h"
#define __global__
#define __constant__
#define SHIM_OPTIX_DEVICE_API
#define SPH_MAX_NUM_COEFFS 16
#define GAUSSIAN_PARTICLE_MAX_ALPHA 0.99f
#define GAUSSIAN_PARTICLE_MIN_KERNEL_DENSITY 0.0113f
#define GAUSSIAN_PARTICLE_MIN_ALPHA (1.0f / 255.0f)
#define GAUSSIAN_PARTICLE_KERNEL_DEGREE PARTICLE_KERNEL_DEGREE
#define PARTICLE_RADIANCE_NUM_COEFFS 16
#define PARTICLE_FEATURE_DIM 48
#define GRT_SLANG_RAYGEN_BUILD
#ifdef REF_SLANG_SH   // the same programs with SH radiance (model.feature_type sh): three ray features, the [n,48] buffer holds the coefficients
#define RAY_FEATURE_DIM 3
#define FEATURE_TRANSFORM_TYPE 0
#else
#define RAY_FEATURE_DIM 24
#define INTERP_POINT_FEATURE_DIM 12
#define FEATURE_TRANSFORM_TYPE 1
#define FEATURE_INTERPOLATION_TYPE 0
#define FEATURE_INTERPOLATION_SUPPORT 1
#define FEATURE_ACTIVATION_TYPE 2
#define FEATURE_ACTIVATION_NUM_FREQUENCIES 1
#endif
#ifdef REF_PRIMITIVE   // a triangle-mesh proxy: -DREF_PRIMITIVE=MOGTracingIcosaHedron (as ref_grt_trace.cpp; tests/golden/grt_trace_nht_mesh.npz)
#define PARTICLE_PRIMITIVE_TYPE MOGPrimitiveTypes::REF_PRIMITIVE
#define SHIM_OPTIX_TRIANGLE_PROXIES
#elif defined(REF_CUSTOM)   // custom primitives: world boxes + the Slang pipeline's own intersection test particleDensityHitCustom (referenceSlangOptix.cu:204-220)
#define PARTICLE_PRIMITIVE_TYPE MOGPrimitiveTypes::MOGTracingCustom
#define SHIM_OPTIX_CUSTOM_PROXIES
#elif defined(REF_SPHERE)   // OptiX's built-in sphere primitive (as ref_grt_trace.cpp)
#define PARTICLE_PRIMITIVE_TYPE MOGPrimitiveTypes::MOGTracingSphere
#define SHIM_OPTIX_SPHERE_PROXIES
#else
#define PARTICLE_PRIMITIVE_TYPE MOGPrimitiveTypes::MOGTracingInstances
#endif
#define PARTICLE_PRIMITIVE_CLAMPED 1
#define ENABLE_HIT_COUNTS
#include "shim/optix.h"
thread_local ShimOptix g_optix;

#include "../_ref/reference_slang_optix_fwd.inc"

#include "ref_grt_emul.inl"

extern "C" {

int ref_grt_slang_ray_feature_dim(void) { return RAY_FEATURE_DIM; }

// as ref_grt_trace_fwd; `features48` is the neural-harmonic feature buffer [n,48], `features` the output [H*W,24]
void ref_grt_trace_slang_fwd(uint32_t n, const float* transforms, const float* density12, const float* features48, int width, int height,
                             const float* ray_to_world, const float* ray_o, const float* ray_d, const float* scene_aabb6, float min_transmittance,
                             float min_response, float min_alpha, float* features, float* density, float* hit_distance2, float* hits_count,
                             int32_t* visibility) {
    set_scene(n, transforms);
    static thread_local std::vector<float> unused_normals;
    unused_normals.assign((size_t)width * height * 3, 0.f);
    set_common_params(width, height, ray_to_world, ray_o, ray_d, density12, features48, scene_aabb6, min_transmittance, min_response, min_alpha, 3,
                      features, density, hit_distance2, unused_normals.data(), hits_count, visibility);
    launch_raygen(width, height);
}

#ifdef REF_PRIMITIVE
// the same programs over the particles' triangle meshes (vertices / triangles as the reference's mesh kernel wrote them, ref_grt_proxies.cpp)
void ref_grt_trace_slang_fwd_mesh(uint32_t n, uint32_t triangles_per_particle, const float* vertices, const int32_t* triangles, const float* density12,
                                  const float* features48, int width, int height, const float* ray_to_world, const float* ray_o, const float* ray_d,
                                  const float* scene_aabb6, float min_transmittance, float min_response, float min_alpha, float* features, float* density,
                                  float* hit_distance2, float* hits_count, int32_t* visibility) {
    static thread_local std::vector<float> unused_normals;
    unused_normals.assign((size_t)width * height * 3, 0.f);
    set_common_params(width, height, ray_to_world, ray_o, ray_d, density12, features48, scene_aabb6, min_transmittance, min_response, min_alpha, 3,
                      features, density, hit_distance2, unused_normals.data(), hits_count, visibility);
    set_scene_triangles(n * triangles_per_particle, triangles_per_particle, vertices, triangles);
    launch_raygen(width, height);
}
#endif

#ifdef REF_CUSTOM
// the same programs over the particles' world boxes (boxes [n,6] as the reference's AABB kernel wrote them, ref_grt_proxies.cpp)
void ref_grt_trace_slang_fwd_custom(uint32_t n, const float* boxes, const float* density12, const float* features48, int width, int height, const float* ray_to_world,
                                    const float* ray_o, const float* ray_d, const float* scene_aabb6, float min_transmittance, float min_response, float min_alpha,
                                    float* features, float* density, float* hit_distance2, float* hits_count, int32_t* visibility) {
    static thread_local std::vector<float> unused_normals;
    unused_normals.assign((size_t)width * height * 3, 0.f);
    set_common_params(width, height, ray_to_world, ray_o, ray_d, density12, features48, scene_aabb6, min_transmittance, min_response, min_alpha, 3,
                      features, density, hit_distance2, unused_normals.data(), hits_count, visibility);
    set_scene_boxes(n, boxes);
    launch_raygen(width, height);
}
#endif

#ifdef REF_SPHERE
// the same programs over the particles' enclosing spheres (centers / radii as the reference's sphere kernel wrote them, ref_grt_proxies.cpp)
void ref_grt_trace_slang_fwd_sphere(uint32_t n, const float* centers, const float* radii, const float* density12, const float* features48, int width, int height,
                                    const float* ray_to_world, const float* ray_o, const float* ray_d, const float* scene_aabb6, float min_transmittance,
                                    float min_response, float min_alpha, float* features, float* density, float* hit_distance2, float* hits_count,
                                    int32_t* visibility) {
    static thread_local std::vector<float> unused_normals;
    unused_normals.assign((size_t)width * height * 3, 0.f);
    set_common_params(width, height, ray_to_world, ray_o, ray_d, density12, features48, scene_aabb6, min_transmittance, min_response, min_alpha, 3,
                      features, density, hit_distance2, unused_normals.data(), hits_count, visibility);
    set_scene_spheres(n, centers, radii);
    launch_raygen(width, height);
}
#endif

}  // extern "C"
