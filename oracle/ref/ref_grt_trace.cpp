// ref_grt_trace.cpp — runs the reference's OWN 3DGRT forward programs on the host: __raygen__rg, trace(), __intersection__is
// and __anyhit__ah of threedgrt_tracer/src/kernels/cuda/referenceOptix.cu, included as they lie (with gaussianParticles.cuh
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

#include "../_ref/reference_optix_fwd.inc"

#include "ref_grt_emul.inl"

extern "C" {

void ref_grt_set_box_test_uses_shrunk_tmax(int on) { g_scene.box_test_uses_shrunk_tmax = on != 0; }

int ref_grt_degree(void) { return PARTICLE_KERNEL_DEGREE; }

// transforms: [n,12] instance matrices of the reference's instance kernel; rays in the space of ray_to_world [12]; outputs [H*W, c]
void ref_grt_trace_fwd(uint32_t n, const float* transforms, const float* density12, const float* sph48, int width, int height,
                       const float* ray_to_world, const float* ray_o, const float* ray_d, const float* scene_aabb6, float min_transmittance,
                       float min_response, float min_alpha, unsigned sph_degree, float* features, float* density, float* hit_distance2,
                       float* normals, float* hits_count, int32_t* visibility) {
    set_scene(n, transforms);
    set_common_params(width, height, ray_to_world, ray_o, ray_d, density12, sph48, scene_aabb6, min_transmittance, min_response, min_alpha,
                      sph_degree, features, density, hit_distance2, normals, hits_count, visibility);
    launch_raygen(width, height);
}

#ifdef REF_PRIMITIVE
// the same programs over the particles' triangle meshes (vertices / triangles as the reference's mesh kernel wrote them, ref_grt_proxies.cpp)
void ref_grt_trace_fwd_mesh(uint32_t n, uint32_t triangles_per_particle, const float* vertices, const int32_t* triangles, const float* density12,
                            const float* sph48, int width, int height, const float* ray_to_world, const float* ray_o, const float* ray_d,
                            const float* scene_aabb6, float min_transmittance, float min_response, float min_alpha, unsigned sph_degree, float* features,
                            float* density, float* hit_distance2, float* normals, float* hits_count, int32_t* visibility) {
    set_common_params(width, height, ray_to_world, ray_o, ray_d, density12, sph48, scene_aabb6, min_transmittance, min_response, min_alpha,
                      sph_degree, features, density, hit_distance2, normals, hits_count, visibility);
    set_scene_triangles(n * triangles_per_particle, triangles_per_particle, vertices, triangles);
    launch_raygen(width, height);
}
#endif

#ifdef REF_CUSTOM
// the same programs over the particles' world boxes (boxes [n,6] as the reference's AABB kernel wrote them, ref_grt_proxies.cpp)
void ref_grt_trace_fwd_custom(uint32_t n, const float* boxes, const float* density12, const float* sph48, int width, int height, const float* ray_to_world,
                              const float* ray_o, const float* ray_d, const float* scene_aabb6, float min_transmittance, float min_response, float min_alpha,
                              unsigned sph_degree, float* features, float* density, float* hit_distance2, float* normals, float* hits_count, int32_t* visibility) {
    set_common_params(width, height, ray_to_world, ray_o, ray_d, density12, sph48, scene_aabb6, min_transmittance, min_response, min_alpha,
                      sph_degree, features, density, hit_distance2, normals, hits_count, visibility);
    set_scene_boxes(n, boxes);
    launch_raygen(width, height);
}
#endif

#ifdef REF_SPHERE
// the same programs over the particles' enclosing spheres (centers [n,3], radii [n] as the reference's sphere kernel wrote them, ref_grt_proxies.cpp)
void ref_grt_trace_fwd_sphere(uint32_t n, const float* centers, const float* radii, const float* density12, const float* sph48, int width, int height,
                              const float* ray_to_world, const float* ray_o, const float* ray_d, const float* scene_aabb6, float min_transmittance, float min_response,
                              float min_alpha, unsigned sph_degree, float* features, float* density, float* hit_distance2, float* normals, float* hits_count,
                              int32_t* visibility) {
    set_common_params(width, height, ray_to_world, ray_o, ray_d, density12, sph48, scene_aabb6, min_transmittance, min_response, min_alpha,
                      sph_degree, features, density, hit_distance2, normals, hits_count, visibility);
    set_scene_spheres(n, centers, radii);
    launch_raygen(width, height);
}
#endif

}  // extern "C"
