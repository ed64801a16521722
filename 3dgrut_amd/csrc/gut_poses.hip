// gut_poses.hip — frame poses from camera-to-world matrices that live in device memory (GutFrame::device_T_to_world): no host
// round trip, no stream synchronisation per frame (the reference's plugin calls `.cpu()` on the pose every iteration,
// threedgut_tracer/tracer.py:413).
//
// This translation unit is built with -ffp-contract=OFF (3dgrut_amd/build.py: FILE_FLAGS): the reference derives the poses on the
// HOST — numpy / torch in tracer.py:359-423, then sensors.h:44-73 compiled by the host compiler, where nothing contracts — and the
// depth keys, i.e. the compositing order, hang on their bits.  With every multiply and add rounded separately the device path
// reproduces the host path bit for bit (tests/test_gut_gpu.py::test_device_poses_equal_host_poses_bit_for_bit).
#include "gut_internal.hpp"

namespace grut {
namespace {

// One wave.  Everything here is a chain of dependent float64 operations (a float64 division alone is ~25 dependent instructions), so the
// kernel's time is its longest chain, not its work: lanes 0-7 take (pose, column of the inverse) - each factors its pose's matrix
// (redundantly: no extra latency) and solves ONE column - then lanes 0 and 4 collect their pose's columns and lane 0 builds the block.
// Same functions, same operation order per value as the sequential host twin (c2w_to_world_to_sensor): identical bits.
__global__ __launch_bounds__(64) void gut_frame_poses_kernel(const float* __restrict__ T_start, const float* __restrict__ T_end, FramePoses* __restrict__ out) {
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x, pose = (lane >> 2) & 1, col = lane & 3;
    const float* m = (pose && T_end) ? T_end : T_start;
    const PoseLU f = pose_lu_factor(m);
    double x[4];
    pose_lu_solve_column(f, col, x);
    const float xf[3] = {(float)x[0], (float)x[1], (float)x[2]};
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int base = lane & 4;   // first lane of this lane's pose
        R[3 * i + 0] = __shfl(xf[i], base + 0, 64);
        R[3 * i + 1] = __shfl(xf[i], base + 1, 64);
        R[3 * i + 2] = __shfl(xf[i], base + 2, 64);
        t[i] = __shfl(xf[i], base + 3, 64);
    }
    float tq[7];
    pose_tquat_from_rows(R, t, tq);
    float ps[7], pe[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        ps[k] = __shfl(tq[k], 0, 64);
        pe[k] = __shfl(tq[k], 4, 64);
    }
    if (lane == 0) *out = make_frame_poses(ps, pe);
}

}  // namespace

void launch_frame_poses(hipStream_t s, const float* T_start, const float* T_end, FramePoses* out) {
    hipLaunchKernelGGL(gut_frame_poses_kernel, dim3(1), dim3(64), 0, s, T_start, T_end, out);
}

}  // namespace grut

// Host twin of the kernel above (same inline code, host compiler): the sensor pose [t, q(xyzw)] the library derives from a
// camera-to-world matrix - for the CPU tests that pin it against tests/golden/pose.npz (made by the reference's own Python).
extern "C" int grut_debug_pose_from_c2w(const float* c2w16, float* out7) {
    if (!c2w16 || !out7) return GRUT_ERR_BAD_INPUT;
    grut::c2w_to_world_to_sensor(c2w16, out7);
    return GRUT_OK;
}

// The complete per-frame pose block (47 floats: start R / t / q, end t / q, mid-exposure view R / t, sensor->world R / t) derived
// from camera-to-world matrices, either by the device kernel (on_device != 0: the matrices are DEVICE pointers, the call
// synchronises the stream) or by its host twin (HOST pointers) - the GPU tests compare the two bit for bit.
extern "C" int grut_debug_frame_poses(void* stream, int on_device, const float* T_start, const float* T_end, float* host_out47) {
    static_assert(sizeof(grut::FramePoses) == 47 * sizeof(float), "FramePoses layout");
    if (!T_start || !host_out47) return GRUT_ERR_BAD_INPUT;
    if (!on_device) {
        float ps[7], pe[7];
        grut::c2w_to_world_to_sensor(T_start, ps);
        grut::c2w_to_world_to_sensor(T_end ? T_end : T_start, pe);
        const grut::FramePoses f = grut::make_frame_poses(ps, pe);
        memcpy(host_out47, &f, sizeof(f));
        return GRUT_OK;
    }
    hipStream_t s = (hipStream_t)stream;
    grut::FramePoses* d = nullptr;
    GRUT_HIP(hipMalloc(&d, sizeof(grut::FramePoses)));
    grut::launch_frame_poses(s, T_start, T_end, d);
    hipError_t e = hipMemcpyAsync(host_out47, d, sizeof(grut::FramePoses), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    GRUT_HIP(e);
    return GRUT_OK;
}
