// ref_camera.cpp — runs the reference's OWN camera projection code on the host:
//   threedgut_tracer/include/3dgut/kernels/cuda/sensors/cameraProjections.cuh  (projectPoint for the OpenCV pinhole / OpenCV
//   fisheye / F-theta models, relativeShutterTime, projectPointWithShutter) and 3dgut/sensors/sensors.h (sensorPoseInverse,
//   interpolatedSensorPose), compiled where they lie with g++, shim/cuda_shim.h and the tiny-cuda-nn stand-in of
//   shim/tiny-cuda-nn/ (that dependency is an empty, un-pinned submodule of the reference).
// TEST INFRASTRUCTURE ONLY: pins the oracle's camera / pose functions (tests/golden/camera.npz, tests/test_oracle_cpu.py).
#include <math.h>
#include "shim/cuda_shim.h"
#include <3dgut/kernels/cuda/sensors/cameraProjections.cuh>

using namespace threedgut;

static TSensorPose pose_from(const float* p) {
    TSensorPose r;
    for (int i = 0; i < 7; ++i) r[i] = p[i];
    return r;
}

extern "C" {

// model: 0 pinhole, 1 fisheye, 2 ftheta; shutter: 0..4 (CameraModelParameters::ShutterType order)
// params layout (floats): principal[2], focal[2], radial[6], tangential[2], thin_prism[4], max_angle, ftheta_reference_poly (0/1),
//                         pixeldist_to_angle[6], angle_to_pixeldist[6], linear_cde[3]      = 33 floats
int ref_project_point_with_shutter(int model, int shutter, int width, int height, const float* prm, const float* pose_start7,
                                   const float* pose_end7, int n_iter, const float* pos3, float tolerance, float* out2) {
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
    st.startPose = pose_from(pose_start7);
    st.endPose = pose_from(pose_end7);
    const tcnn::ivec2 res(width, height);
    const tcnn::vec3 p(pos3[0], pos3[1], pos3[2]);
    tcnn::vec2 out = tcnn::vec2::zero();
    bool ok;
    switch (n_iter) {
    case 0: ok = projectPointWithShutter<0>(p, res, m, st, tolerance, out); break;
    case 2: ok = projectPointWithShutter<2>(p, res, m, st, tolerance, out); break;
    default: ok = projectPointWithShutter<5>(p, res, m, st, tolerance, out); break;
    }
    out2[0] = out.x; out2[1] = out.y;
    return ok ? 1 : 0;
}

void ref_pose_inverse(const float* p7, float* out7) {
    const TSensorPose r = sensorPoseInverse(pose_from(p7));
    for (int i = 0; i < 7; ++i) out7[i] = r[i];
}
void ref_pose_interpolate(const float* a7, const float* b7, float t, float* out7) {
    const TSensorPose r = interpolatedSensorPose(pose_from(a7), pose_from(b7), t);
    for (int i = 0; i < 7; ++i) out7[i] = r[i];
}

}  // extern "C"
