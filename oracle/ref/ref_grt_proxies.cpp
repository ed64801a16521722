// ref_grt_proxies.cpp — runs the reference's OWN proxy-geometry kernels on the host: kernelScale, computeGaussianEnclosing-
// AABBKernel and computeGaussianEnclosingInstancesKernel of threedgrt_tracer/src/particlePrimitives.cu (:27-51, :498-610).
// The Makefile cuts the device part of that file (everything before its first host helper, which starts the torch / <<< >>>
// launch wrappers) into oracle/_ref/particle_primitives_kernels.inc — git-ignored, never committed — and this file calls
// the kernels thread by thread.  TEST INFRASTRUCTURE ONLY: pins oracle/grt_oracle.c (tests/golden/grt_proxies.npz).
#include <math.h>
#include <vector>
#include "shim/cuda_shim.h"
#define __global__
struct ShimDim3 { unsigned x, y, z; };
static thread_local ShimDim3 blockIdx = {0, 0, 0}, blockDim = {1, 1, 1}, threadIdx = {0, 0, 0};
template <class T> static inline T atomicAdd(T* a, T v) { T o = *a; *a += v; return o; }
#include "../_ref/particle_primitives_kernels.inc"
}  // namespace  (the cut ends inside the file's anonymous namespace)

extern "C" {
float ref_kernel_scale(float density, float min_response, unsigned opts, float degree) { return kernelScale(density, min_response, opts, degree); }
// pos [n,3], rot [n,4] wxyz, scl [n,3], dns [n] -> aabb [n,6] (min xyz, max xyz), transform [n,12] (row-major 3x4 instance matrix)
void ref_enclosing_proxies(unsigned n, const float* pos, const float* rot, const float* scl, const float* dns, float min_response, unsigned opts,
                           float degree, float* aabb, float* transform) {
    OptixAabb* boxes = new OptixAabb[n];
    OptixInstance* inst = new OptixInstance[n];
    blockDim.x = 1; threadIdx.x = 0;
    for (unsigned i = 0; i < n; ++i) {
        blockIdx.x = i;
        computeGaussianEnclosingAABBKernel(n, (const float3*)pos, (const float4*)rot, (const float3*)scl, dns, min_response, opts, degree, boxes);
        computeGaussianEnclosingInstancesKernel(n, (const float3*)pos, (const float4*)rot, (const float3*)scl, dns, min_response, opts, degree, 0, inst);
    }
    for (unsigned i = 0; i < n; ++i) {
        aabb[6 * i + 0] = boxes[i].minX; aabb[6 * i + 1] = boxes[i].minY; aabb[6 * i + 2] = boxes[i].minZ;
        aabb[6 * i + 3] = boxes[i].maxX; aabb[6 * i + 4] = boxes[i].maxY; aabb[6 * i + 5] = boxes[i].maxZ;
        for (int k = 0; k < 12; ++k) transform[12 * i + k] = inst[i].transform[k];
    }
    delete[] boxes;
    delete[] inst;
}

// render.primitive_type sphere: centers [n,3] and radii [n] of the particles' enclosing spheres, written by the reference's own kernel
// (computeGaussianEnclosingSphereKernel, particlePrimitives.cu:386-403)
void ref_enclosing_spheres(unsigned n, const float* pos, const float* rot, const float* scl, const float* dns, float min_response, unsigned opts, float degree,
                           float* centers, float* radii) {
    blockDim.x = 1; threadIdx.x = 0;
    for (unsigned i = 0; i < n; ++i) {
        blockIdx.x = i;
        computeGaussianEnclosingSphereKernel(n, (const float3*)pos, (const float4*)rot, (const float3*)scl, dns, min_response, opts, degree, (float3*)centers, radii);
    }
}

// the trisurfel meshes together with the per-particle {normal, density} rows the surfel pipeline reads (params.particleExtendedData,
// optixTracer.cpp:735-748, 937): vertices [n * 4, 3], triangles [n * 2, 3], normal_density [n, 4]
void ref_enclosing_trisurfels(unsigned n, const float* pos, const float* rot, const float* scl, const float* dns, float min_response, unsigned opts, float degree,
                              float* vertices, int* triangles, float* normal_density) {
    blockDim.x = 1; threadIdx.x = 0;
    for (unsigned i = 0; i < n; ++i) {
        blockIdx.x = i;
        computeGaussianEnclosingTriSurfelKernel<false>(n, (const float3*)pos, (const float4*)rot, (const float3*)scl, dns, min_response, opts, degree, (float3*)vertices,
                                                       (int3*)triangles, (float4*)normal_density);
    }
}

// the triangle-mesh proxies: prim 1 icosahedron, 2 octahedron, 3 tetrahedron, 4 diamond (GRUT_PRIM_*), 6 trisurfel, 7 trihexa (checker only) -> vertices [n * V, 3] in world space,
// triangles [n * T, 3] (indices into all vertices), written by the reference's own mesh kernel of that type.  Returns T (V through *num_vertices).
unsigned ref_enclosing_mesh(int prim, unsigned n, const float* pos, const float* rot, const float* scl, const float* dns, float min_response, unsigned opts,
                            float degree, float* vertices, int* triangles, unsigned* num_vertices) {
    blockDim.x = 1; threadIdx.x = 0;
    std::vector<float4> normal_density(n);   // (the trisurfel kernel's per-particle normal / density: read by the surfel pipeline only)
    for (unsigned i = 0; i < n; ++i) {
        blockIdx.x = i;
        switch (prim) {
        case 1: computeGaussianEnclosingIcosaHedronKernel(n, (const float3*)pos, (const float4*)rot, (const float3*)scl, dns, min_response, opts, degree, (float3*)vertices, (int3*)triangles); break;
        case 2: computeGaussianEnclosingOctaHedronKernel(n, (const float3*)pos, (const float4*)rot, (const float3*)scl, dns, min_response, opts, degree, (float3*)vertices, (int3*)triangles); break;
        case 3: computeGaussianEnclosingTetraHedronKernel(n, (const float3*)pos, (const float4*)rot, (const float3*)scl, dns, min_response, opts, degree, (float3*)vertices, (int3*)triangles); break;
        case 4: computeGaussianEnclosingDiamondKernel(n, (const float3*)pos, (const float4*)rot, (const float3*)scl, dns, min_response, opts, degree, (float3*)vertices, (int3*)triangles); break;
        case 7: computeGaussianEnclosingTriHexaKernel(n, (const float3*)pos, (const float4*)rot, (const float3*)scl, dns, min_response, opts, degree, (float3*)vertices, (int3*)triangles); break;
        case 6: computeGaussianEnclosingTriSurfelKernel<false>(n, (const float3*)pos, (const float4*)rot, (const float3*)scl, dns, min_response, opts, degree, (float3*)vertices, (int3*)triangles, normal_density.data()); break;
        default: return 0;
        }
    }
    const unsigned nv[8] = {0, icosaHedronNumVrt, octaHedronNumVrt, tetraHedronNumVrt, diamondNumVrt, 0, triSurfelNumVrt, triHexaNumVrt};
    const unsigned nt[8] = {0, icosaHedronNumTri, octaHedronNumTri, tetraHedronNumTri, diamondNumTri, 0, triSurfelNumTri, triHexaNumTri};
    *num_vertices = nv[prim];
    return nt[prim];
}
}
