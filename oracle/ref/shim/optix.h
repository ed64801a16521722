// Stand-in for the few OptiX 7.5 host-API types that the reference's device headers / proxy kernels mention
// (OptiX is an un-vendored dependency of the reference: threedgrt_tracer/dependencies/optix-dev is empty).
// Layouts follow the public OptiX headers.  TEST INFRASTRUCTURE ONLY; contains no reference code.
#pragma once
#include <cstdint>
typedef unsigned long long OptixTraversableHandle;
struct OptixAabb { float minX, minY, minZ, maxX, maxY, maxZ; };
struct OptixInstance {
    float transform[12];
    unsigned int instanceId, sbtOffset, visibilityMask, flags;
    OptixTraversableHandle traversableHandle;
    unsigned int pad[2];
};
enum { OPTIX_INSTANCE_FLAG_NONE = 0 };

// ---- device API, emulated on the host (oracle/ref/ref_grt_trace.cpp) ---------------------------------------------------
// State of the one ray being traced and of the instance whose programs are running.  The traversal itself — which
// instances a ray is offered, in which order, against which interval — is OptiX's and is restated in ref_grt_trace.cpp;
// everything the reference's raygen / intersection / any-hit programs do runs as written.
#ifdef SHIM_OPTIX_DEVICE_API
struct ShimOptix {
    uint3 launchIndex;
    float3 worldOrigin, worldDirection, objectOrigin, objectDirection;
    float tmin, tmax;            // tmax: the ray's current far end; inside an any-hit program the distance of the reported hit
    unsigned instance;
    uint32_t* payload[32];
    bool ignore;
    // playground (closest-hit programs over a triangle GAS): launch size, the hit triangle, its barycentrics and vertices
    uint3 launchDim;
    unsigned primitive;
    float2 barycentrics;
    float3 triangle[3];
};
extern thread_local ShimOptix g_optix;
typedef unsigned OptixVisibilityMask;
enum { OPTIX_RAY_FLAG_NONE = 0, OPTIX_RAY_FLAG_DISABLE_CLOSESTHIT = 1 << 3, OPTIX_RAY_FLAG_CULL_BACK_FACING_TRIANGLES = 1 << 4 };
inline uint3 optixGetLaunchIndex() { return g_optix.launchIndex; }
inline float optixGetRayTmin() { return g_optix.tmin; }
inline float optixGetRayTmax() { return g_optix.tmax; }
inline unsigned optixGetInstanceIndex() { return g_optix.instance; }
#ifdef SHIM_OPTIX_PLAYGROUND
inline unsigned optixGetPrimitiveIndex() { return g_optix.primitive; }   // the hit triangle (closest-hit) / 0 for the one custom primitive of a particle
inline uint3 optixGetLaunchDimensions() { return g_optix.launchDim; }
inline float2 optixGetTriangleBarycentrics() { return g_optix.barycentrics; }
inline OptixTraversableHandle optixGetGASTraversableHandle() { return 0; }
inline unsigned optixGetSbtGASIndex() { return 0; }
inline void optixGetTriangleVertexData(OptixTraversableHandle, unsigned, unsigned, float, float3 v[3]) { v[0] = g_optix.triangle[0]; v[1] = g_optix.triangle[1]; v[2] = g_optix.triangle[2]; }
enum { OPTIX_RAY_FLAG_DISABLE_ANYHIT = 1 << 0 };
// the two-register trace of the mesh pass (trace.cuh:175-195)
void optixTrace(OptixTraversableHandle handle, float3 origin, float3 direction, float tmin, float tmax, float time, OptixVisibilityMask mask,
                unsigned flags, unsigned sbtOffset, unsigned sbtStride, unsigned missIndex, uint32_t& p0, uint32_t& p1);
#elif defined(SHIM_OPTIX_TRIANGLE_PROXIES) || defined(SHIM_OPTIX_CUSTOM_PROXIES) || defined(SHIM_OPTIX_SPHERE_PROXIES)
inline float2 optixGetTriangleBarycentrics() { return g_optix.barycentrics; }   // of the reported triangle hit: hit = (1 - u - v) V0 + u V1 + v V2 (barycentricSurfelsOptix.cu:210)
inline unsigned optixGetPrimitiveIndex() { return g_optix.primitive; }   // the hit triangle of the particles' triangle GAS (optixTracer.cpp:836-845) / the particle's custom primitive (:810-817)
#else
inline unsigned optixGetPrimitiveIndex() { return 0; }   // the instanced BLAS holds one custom primitive (optixTracer.cpp:551-563)
#endif
inline float3 optixGetObjectRayOrigin() { return g_optix.objectOrigin; }
inline float3 optixGetObjectRayDirection() { return g_optix.objectDirection; }
inline float3 optixGetWorldRayOrigin() { return g_optix.worldOrigin; }
inline float3 optixGetWorldRayDirection() { return g_optix.worldDirection; }
inline void optixIgnoreIntersection() { g_optix.ignore = true; }
bool optixReportIntersection(float t, unsigned kind);
#define SHIM_OPTIX_PAYLOAD(N)                                             \
    inline uint32_t optixGetPayload_##N() { return *g_optix.payload[N]; } \
    inline void optixSetPayload_##N(uint32_t v) { *g_optix.payload[N] = v; }
SHIM_OPTIX_PAYLOAD(0) SHIM_OPTIX_PAYLOAD(1) SHIM_OPTIX_PAYLOAD(2) SHIM_OPTIX_PAYLOAD(3) SHIM_OPTIX_PAYLOAD(4) SHIM_OPTIX_PAYLOAD(5)
SHIM_OPTIX_PAYLOAD(6) SHIM_OPTIX_PAYLOAD(7) SHIM_OPTIX_PAYLOAD(8) SHIM_OPTIX_PAYLOAD(9) SHIM_OPTIX_PAYLOAD(10) SHIM_OPTIX_PAYLOAD(11)
SHIM_OPTIX_PAYLOAD(12) SHIM_OPTIX_PAYLOAD(13) SHIM_OPTIX_PAYLOAD(14) SHIM_OPTIX_PAYLOAD(15) SHIM_OPTIX_PAYLOAD(16) SHIM_OPTIX_PAYLOAD(17)
SHIM_OPTIX_PAYLOAD(18) SHIM_OPTIX_PAYLOAD(19) SHIM_OPTIX_PAYLOAD(20) SHIM_OPTIX_PAYLOAD(21) SHIM_OPTIX_PAYLOAD(22) SHIM_OPTIX_PAYLOAD(23)
SHIM_OPTIX_PAYLOAD(24) SHIM_OPTIX_PAYLOAD(25) SHIM_OPTIX_PAYLOAD(26) SHIM_OPTIX_PAYLOAD(27) SHIM_OPTIX_PAYLOAD(28) SHIM_OPTIX_PAYLOAD(29)
SHIM_OPTIX_PAYLOAD(30) SHIM_OPTIX_PAYLOAD(31)
#undef SHIM_OPTIX_PAYLOAD
void optixTrace(OptixTraversableHandle handle, float3 origin, float3 direction, float tmin, float tmax, float time, OptixVisibilityMask mask,
                unsigned flags, unsigned sbtOffset, unsigned sbtStride, unsigned missIndex, uint32_t& p0, uint32_t& p1, uint32_t& p2, uint32_t& p3,
                uint32_t& p4, uint32_t& p5, uint32_t& p6, uint32_t& p7, uint32_t& p8, uint32_t& p9, uint32_t& p10, uint32_t& p11, uint32_t& p12,
                uint32_t& p13, uint32_t& p14, uint32_t& p15, uint32_t& p16, uint32_t& p17, uint32_t& p18, uint32_t& p19, uint32_t& p20,
                uint32_t& p21, uint32_t& p22, uint32_t& p23, uint32_t& p24, uint32_t& p25, uint32_t& p26, uint32_t& p27, uint32_t& p28,
                uint32_t& p29, uint32_t& p30, uint32_t& p31);
// the 30-register trace of barycentricSurfelsOptix.cu:58-68 (ten hits x {particle, distance, squared distance})
inline void optixTrace(OptixTraversableHandle handle, float3 origin, float3 direction, float tmin, float tmax, float time, OptixVisibilityMask mask,
                       unsigned flags, unsigned sbtOffset, unsigned sbtStride, unsigned missIndex, uint32_t& p0, uint32_t& p1, uint32_t& p2, uint32_t& p3,
                       uint32_t& p4, uint32_t& p5, uint32_t& p6, uint32_t& p7, uint32_t& p8, uint32_t& p9, uint32_t& p10, uint32_t& p11, uint32_t& p12,
                       uint32_t& p13, uint32_t& p14, uint32_t& p15, uint32_t& p16, uint32_t& p17, uint32_t& p18, uint32_t& p19, uint32_t& p20,
                       uint32_t& p21, uint32_t& p22, uint32_t& p23, uint32_t& p24, uint32_t& p25, uint32_t& p26, uint32_t& p27, uint32_t& p28,
                       uint32_t& p29) {
    uint32_t unused30 = 0, unused31 = 0;
    optixTrace(handle, origin, direction, tmin, tmax, time, mask, flags, sbtOffset, sbtStride, missIndex, p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11, p12,
               p13, p14, p15, p16, p17, p18, p19, p20, p21, p22, p23, p24, p25, p26, p27, p28, p29, unused30, unused31);
}
#endif
