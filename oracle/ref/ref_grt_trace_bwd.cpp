// ref_grt_trace_bwd.cpp — runs the reference's OWN 3DGRT backward programs on the host: __raygen__rg, trace(), __intersection__is
// and __anyhit__ah of threedgrt_tracer/src/kernels/cuda/referenceBwdOptix.cu, included as they lie (with gaussianParticles.cuh
// and pipelineParameters.h below them), over the proxy instances produced by the reference's own kernel
// (computeGaussianEnclosingInstancesKernel, see ref_grt_proxies.cpp).
//
// What is emulated, i.e. NOT the reference's and not pinned by this library, is OptiX itself (an un-vendored dependency:
// threedgrt_tracer/dependencies/optix-dev is empty): `optixTrace` below offers the ray every instance in index order — the
// instance's inverse transform gives the object-space ray, the instanced BLAS is the single custom primitive with the box
// [-1,1]^3 (optixTracer.cpp:551-563), the intersection program runs when that box overlaps the ray's CURRENT interval
// [tmin, tmax], `optixReportIntersection` runs the any-hit program and shrinks tmax unless the hit is ignored.  OptiX leaves
// the order of instances unspecified; since hits are accepted only once the 16-entry payload is full, the payload ends
// up holding the 16 nearest for any order (up to exact ties and to boxes culled by an already shrunk tmax).
// TEST INFRASTRUCTURE ONLY: pins oracle/grt_oracle.c's trace rounds (tests/golden/grt_trace.npz).
#include <math.h>
#include <type_traits>
#include <vector>
#include "shim/cuda_shim.h"
#define __global__
#define __constant__
#define SHIM_OPTIX_DEVICE_API
#define SPH_MAX_NUM_COEFFS 16
#define GAUSSIAN_PARTICLE_MAX_ALPHA 0.99f
#define PARTICLE_FEATURE_DIM 48
#define RAY_FEATURE_DIM 3
#define FEATURE_TRANSFORM_TYPE 0
#ifdef REF_PRIMITIVE   // a triangle-mesh proxy: -DREF_PRIMITIVE=MOGTracingIcosaHedron ... (optixTracer.cpp:176-201)
#define PARTICLE_PRIMITIVE_TYPE MOGPrimitiveTypes::REF_PRIMITIVE
#define SHIM_OPTIX_TRIANGLE_PROXIES
#elif defined(REF_CUSTOM)   // render.primitive_type custom: world boxes + intersectCustomParticle (optixTracer.cpp:197-198)
#define PARTICLE_PRIMITIVE_TYPE MOGPrimitiveTypes::MOGTracingCustom
#define SHIM_OPTIX_CUSTOM_PROXIES
#elif defined(REF_SPHERE)   // render.primitive_type sphere: OptiX's built-in sphere primitive (optixTracer.cpp:189-190, 765-781)
#define PARTICLE_PRIMITIVE_TYPE MOGPrimitiveTypes::MOGTracingSphere
#define SHIM_OPTIX_SPHERE_PROXIES
#else
#define PARTICLE_PRIMITIVE_TYPE MOGPrimitiveTypes::MOGTracingInstances
#endif
#define PARTICLE_PRIMITIVE_CLAMPED 1
#define ENABLE_NORMALS
#define ENABLE_HIT_COUNTS
#include "shim/optix.h"
thread_local ShimOptix g_optix;

#include "../_ref/reference_optix_bwd.inc"

#include "ref_grt_emul.inl"

extern "C" {

void ref_grt_set_box_test_uses_shrunk_tmax(int on) { g_scene.box_test_uses_shrunk_tmax = on != 0; }

static void bwd_params_and_launch(uint32_t n, const float* density12, const float* sph48, int width, int height, const float* ray_to_world, const float* ray_o,
                                  const float* ray_d, const float* scene_aabb6, float min_transmittance, float min_response, float min_alpha, unsigned sph_degree,
                                  const float* features, const float* density, const float* hit_distance2, const float* g_features, const float* g_density,
                                  const float* g_hit_distance, float* g_density12, float* g_sph48, uint32_t triangles_per_particle, const float* vertices,
                                  const int32_t* triangles) {
    std::vector<float> dummy3((size_t)width * height * 3, 0.f), dummy1((size_t)width * height, 0.f);
    std::vector<int32_t> vis(n, 0);
    set_common_params(width, height, ray_to_world, ray_o, ray_d, density12, sph48, scene_aabb6, min_transmittance, min_response, min_alpha,
                      sph_degree, const_cast<float*>(features), const_cast<float*>(density), const_cast<float*>(hit_distance2), dummy3.data(),
                      dummy1.data(), vis.data());
    if (vertices) set_scene_triangles(n * triangles_per_particle, triangles_per_particle, vertices, triangles);
    if (g_scene.num_spheres) params.gPrimNumTri = 1;   // (set_common_params cleared it; see set_scene_spheres)
    const int32_t sz3[4] = {1, height, width, 3}, st3[4] = {height * width * 3, width * 3, 3, 1};
    const int32_t sz1[4] = {1, height, width, 1}, st1[4] = {height * width, width, 1, 1};
    fill_accessor(params.rayFeaturesGrad, const_cast<float*>(g_features), sz3, st3);
    fill_accessor(params.rayDensityGrad, const_cast<float*>(g_density), sz1, st1);
    fill_accessor(params.rayHitDistanceGrad, const_cast<float*>(g_hit_distance), sz1, st1);
    fill_accessor(params.rayNormalGrad, dummy3.data(), sz3, st3);
    params.particleDensityGrad = reinterpret_cast<ParticleDensity*>(g_density12);
    params.particleFeaturesGrad = g_sph48;
    launch_raygen(width, height);
}

// forward results (features / density / hit_distance2) and upstream gradients in, particle gradients out ([n,12], [n,48],
// zero-filled by the caller like optixTracer.cpp:1010-1031 does)
void ref_grt_trace_bwd(uint32_t n, const float* transforms, const float* density12, const float* sph48, int width, int height,
                       const float* ray_to_world, const float* ray_o, const float* ray_d, const float* scene_aabb6, float min_transmittance,
                       float min_response, float min_alpha, unsigned sph_degree, const float* features, const float* density,
                       const float* hit_distance2, const float* g_features, const float* g_density, const float* g_hit_distance,
                       float* g_density12, float* g_sph48) {
    set_scene(n, transforms);
    bwd_params_and_launch(n, density12, sph48, width, height, ray_to_world, ray_o, ray_d, scene_aabb6, min_transmittance, min_response, min_alpha, sph_degree,
                          features, density, hit_distance2, g_features, g_density, g_hit_distance, g_density12, g_sph48, 0, nullptr, nullptr);
}

#ifdef REF_PRIMITIVE
void ref_grt_trace_bwd_mesh(uint32_t n, uint32_t triangles_per_particle, const float* vertices, const int32_t* triangles, const float* density12,
                            const float* sph48, int width, int height, const float* ray_to_world, const float* ray_o, const float* ray_d,
                            const float* scene_aabb6, float min_transmittance, float min_response, float min_alpha, unsigned sph_degree,
                            const float* features, const float* density, const float* hit_distance2, const float* g_features, const float* g_density,
                            const float* g_hit_distance, float* g_density12, float* g_sph48) {
    bwd_params_and_launch(n, density12, sph48, width, height, ray_to_world, ray_o, ray_d, scene_aabb6, min_transmittance, min_response, min_alpha, sph_degree,
                          features, density, hit_distance2, g_features, g_density, g_hit_distance, g_density12, g_sph48, triangles_per_particle, vertices,
                          triangles);
}
#endif

#ifdef REF_SPHERE
void ref_grt_trace_bwd_sphere(uint32_t n, const float* centers, const float* radii, const float* density12, const float* sph48, int width, int height,
                              const float* ray_to_world, const float* ray_o, const float* ray_d, const float* scene_aabb6, float min_transmittance, float min_response,
                              float min_alpha, unsigned sph_degree, const float* features, const float* density, const float* hit_distance2, const float* g_features,
                              const float* g_density, const float* g_hit_distance, float* g_density12, float* g_sph48) {
    set_scene_spheres(n, centers, radii);
    bwd_params_and_launch(n, density12, sph48, width, height, ray_to_world, ray_o, ray_d, scene_aabb6, min_transmittance, min_response, min_alpha, sph_degree,
                          features, density, hit_distance2, g_features, g_density, g_hit_distance, g_density12, g_sph48, 0, nullptr, nullptr);
}
#endif

#ifdef REF_CUSTOM
void ref_grt_trace_bwd_custom(uint32_t n, const float* boxes, const float* density12, const float* sph48, int width, int height, const float* ray_to_world,
                              const float* ray_o, const float* ray_d, const float* scene_aabb6, float min_transmittance, float min_response, float min_alpha,
                              unsigned sph_degree, const float* features, const float* density, const float* hit_distance2, const float* g_features,
                              const float* g_density, const float* g_hit_distance, float* g_density12, float* g_sph48) {
    set_scene_boxes(n, boxes);
    bwd_params_and_launch(n, density12, sph48, width, height, ray_to_world, ray_o, ray_d, scene_aabb6, min_transmittance, min_response, min_alpha, sph_degree,
                          features, density, hit_distance2, g_features, g_density, g_hit_distance, g_density12, g_sph48, 0, nullptr, nullptr);
}
#endif

}  // extern "C"
