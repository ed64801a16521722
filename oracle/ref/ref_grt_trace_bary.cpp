// ref_grt_trace_bary.cpp - runs the reference's SURFEL forward pipeline on the host: __raygen__rg, trace() and __anyhit__ah of
// threedgrt_tracer/src/kernels/cuda/barycentricSurfelsOptix.cu (render.pipeline_type barycentricSurfels), included as they lie, over
// the trisurfel meshes and the per-particle {normal, density} rows of the reference's own kernel (computeGaussianEnclosingTriSurfelKernel,
// particlePrimitives.cu:155-205 through ref_grt_proxies.cpp).  OptiX itself is the emulation of ref_grt_emul.inl: built-in triangles
// WITHOUT face culling (PipelineParameters::SurfelPrimitive -> OPTIX_RAY_FLAG_NONE, :62), the any-hit program is handed the hit
// triangle's barycentrics.  The reference has no backward program for this pipeline (no barycentricSurfelsBwdOptix.cu in the checkout).
// TEST INFRASTRUCTURE ONLY: pins oracle/grt_oracle.c's orc_grt_trace_bary_fwd (tests/golden/grt_trace_bary.npz).
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
#define PARTICLE_PRIMITIVE_TYPE MOGPrimitiveTypes::MOGTracingTriSurfel
#define SHIM_OPTIX_TRIANGLE_PROXIES
#define SHIM_OPTIX_NO_INTERSECTION_PROGRAM
#define PARTICLE_PRIMITIVE_CLAMPED 1
#define ENABLE_NORMALS
#define ENABLE_HIT_COUNTS
#include "shim/optix.h"
thread_local ShimOptix g_optix;

#include "../_ref/barycentric_surfels_optix.inc"

#include "ref_grt_emul.inl"

extern "C" {

int ref_grt_degree(void) { return PARTICLE_KERNEL_DEGREE; }

// vertices / triangles / normal_density [n,4]: as the reference's trisurfel kernel wrote them (ref_enclosing_trisurfels); outputs [H*W, c]
void ref_grt_trace_bary_fwd(uint32_t n, const float* vertices, const int32_t* triangles, const float* normal_density4, const float* density12, const float* sph48,
                            int width, int height, const float* ray_to_world, const float* ray_o, const float* ray_d, const float* scene_aabb6,
                            float min_transmittance, float min_response, float min_alpha, unsigned sph_degree, float* features, float* density,
                            float* hit_distance2, float* normals, float* hits_count, int32_t* visibility) {
    set_common_params(width, height, ray_to_world, ray_o, ray_d, density12, sph48, scene_aabb6, min_transmittance, min_response, min_alpha,
                      sph_degree, features, density, hit_distance2, normals, hits_count, visibility);
    set_scene_triangles(n * 2, 2, vertices, triangles);
    params.particleExtendedData = normal_density4;
    launch_raygen(width, height);
}

}  // extern "C"
