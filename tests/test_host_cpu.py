"""CPU: the host side of the plugin boundary — pose / camera marshalling, configuration parsing, the drop-in package names,
and the refusal to run without the HIP path (reference behaviour cited per test)."""
import importlib
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
camera = importlib.import_module("3dgrut_amd.camera")
abi = importlib.import_module("3dgrut_amd._abi")
syn = importlib.import_module("workloads.synthetic")


def _rot_from_xyzw(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_pose_marshalling_inverts_the_camera_to_world_matrix():
    """tracer.py:413-423, 359-380: the plugin ships inv(T_to_world) as [t, q(x,y,z,w)]; every branch of the matrix -> quaternion
    conversion (largest of the three diagonal entries / trace, tracer.py:88-136) is exercised by rotations about each axis."""
    rng = np.random.default_rng(4)
    cases = [syn.orbit_pose(i, n_views=7) for i in range(7)]
    for axis in range(3):          # half turns about x, y, z: the trace is -1 and a diagonal entry wins
        R = -np.eye(3)
        R[axis, axis] = 1.0
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = rng.normal(size=3)
        cases.append(T)
    for _ in range(20):
        A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        if np.linalg.det(A) < 0:
            A[:, 0] = -A[:, 0]
        T = np.eye(4)
        T[:3, :3] = A
        T[:3, 3] = rng.normal(size=3) * 3
        cases.append(T)
    for T in cases:
        tq = camera.world_to_sensor_tquat(np.asarray(T, np.float32))
        assert tq.dtype == np.float32 and tq.shape == (7,)
        w2c = np.linalg.inv(np.asarray(T, np.float64))
        assert abs(np.linalg.norm(tq[3:]) - 1.0) < 1e-6
        assert np.abs(_rot_from_xyzw(tq[3:]) - w2c[:3, :3]).max() < 2e-6
        assert np.abs(tq[:3] - w2c[:3, 3]).max() < 1e-5


def test_sensor_poses_equal_the_reference_plugins_bit_for_bit(grut_lib):
    """tests/golden/pose.npz holds what the REFERENCE's own Python (Tracer.__create_camera_parameters executed from the checkout,
    tests/golden/make_pose_golden.py) makes of 1033 start / end camera-to-world pairs: float64 np.linalg.inv, one rounding, float32 torch
    quaternion.  Both of this library's derivations - the plugin's host code (camera.py) and the host twin of the device code
    (csrc/camera.hpp: c2w_to_world_to_sensor, float64 LU inverse) - must return the SAME BITS: the pose's low bits reach the depth keys
    and with them the compositing order (round 3 derived the quaternion in float64 on the host and inverted in float32 on the device:
    929 of 1033 poses differed from the reference's in the last bits)."""
    import ctypes as C
    g = np.load(os.path.join(ROOT, "tests", "golden", "pose.npz"))
    fp = C.POINTER(C.c_float)
    for c2w, want in ((g["c2w"], g["tquat_start"]), (g["c2w_end"], g["tquat_end"])):
        host = np.stack([camera.world_to_sensor_tquat(m) for m in c2w])
        assert host.dtype == np.float32 and np.array_equal(host.view(np.uint32), want.view(np.uint32))
        twin = np.zeros_like(want)
        for i, m in enumerate(c2w):
            m = np.ascontiguousarray(m, np.float32)
            assert grut_lib.grut_debug_pose_from_c2w(m.ctypes.data_as(fp), twin[i].ctypes.data_as(fp)) == 0
        assert np.array_equal(twin.view(np.uint32), want.view(np.uint32)), int((twin.view(np.uint32) != want.view(np.uint32)).any(1).sum())
    # the pose block the kernels read, host twin: the same call on the pairs as 4x4 matrices (used by the GPU test as its reference)
    out = np.zeros(47, np.float32)
    a, b = np.ascontiguousarray(g["c2w"][3], np.float32), np.ascontiguousarray(g["c2w_end"][3], np.float32)
    assert grut_lib.grut_debug_frame_poses(None, 0, a.ctypes.data, b.ctypes.data, out.ctypes.data_as(fp)) == 0
    assert np.array_equal(out[9:12].view(np.uint32), g["tquat_start"][3, :3].view(np.uint32))      # start_t
    assert np.array_equal(out[12:16].view(np.uint32), g["tquat_start"][3, 3:].view(np.uint32))     # start_q
    assert np.array_equal(out[16:19].view(np.uint32), g["tquat_end"][3, :3].view(np.uint32))       # end_t


def _batch(**kw):
    H, W = 6, 8
    base = dict(rays_ori=np.zeros((1, H, W, 3), np.float32), rays_dir=np.zeros((1, H, W, 3), np.float32), T_to_world=syn.orbit_pose(2)[None])
    base.update(kw)
    return base


def test_plain_intrinsics_become_a_centred_pinhole():
    """tracer.py:426-444: [fx, fy, cx, cy] -> OpenCV pinhole, resolution (int(2cx), int(2cy)), principal point at its centre,
    focal lengths through the field of view, zero distortion, global shutter; start and end pose identical."""
    cam, ps, pe = camera.camera_from_batch(_batch(intrinsics=[100.0, 110.0, 40.3, 30.2]))
    assert cam.model == abi.CAMERA_OPENCV_PINHOLE and cam.shutter == abi.SHUTTER_GLOBAL
    assert (cam.width, cam.height) == (80, 60)
    assert list(cam.principal_point) == [40.0, 30.0]
    assert abs(cam.focal_length[0] - 100.0) < 1e-4 and abs(cam.focal_length[1] - 110.0) < 1e-4
    assert not any(cam.radial) and not any(cam.tangential) and not any(cam.thin_prism)
    assert np.array_equal(ps, pe)
    # dict and attribute batches are the same thing to the plugin
    cam2, ps2, _ = camera.camera_from_batch(SimpleNamespace(**_batch(intrinsics=[100.0, 110.0, 40.3, 30.2])))
    assert bytes(cam2) == bytes(cam) and np.array_equal(ps, ps2)


def test_camera_model_dicts_and_rolling_shutter_poses():
    """tracer.py:446-488: the three *CameraModelParameters dicts; shutter types by name or enum; T_to_world_end gives the end pose."""
    end = syn.orbit_pose(3)[None]
    pin = dict(resolution=np.array([64, 48], np.uint32), shutter_type="ROLLING_BOTTOM_TO_TOP", principal_point=np.array([31.0, 23.5], np.float32),
               focal_length=np.array([70.0, 71.0], np.float32), radial_coeffs=np.arange(6, dtype=np.float32) * 0.01,
               tangential_coeffs=np.array([0.001, -0.002], np.float32), thin_prism_coeffs=np.array([1e-3, 2e-3, 3e-3, 4e-3], np.float32))
    cam, ps, pe = camera.camera_from_batch(_batch(intrinsics_OpenCVPinholeCameraModelParameters=pin, T_to_world_end=end))
    assert cam.model == abi.CAMERA_OPENCV_PINHOLE and cam.shutter == abi.SHUTTER_ROLLING_BOTTOM_TO_TOP and (cam.width, cam.height) == (64, 48)
    assert np.allclose(list(cam.radial), pin["radial_coeffs"]) and np.allclose(list(cam.thin_prism), pin["thin_prism_coeffs"])
    assert not np.array_equal(ps, pe) and np.array_equal(pe, camera.world_to_sensor_tquat(end[0]))
    fish = dict(resolution=np.array([64, 48], np.uint32), shutter_type=SimpleNamespace(name="GLOBAL"), principal_point=np.array([32.0, 24.0], np.float32),
                focal_length=np.array([30.0, 30.0], np.float32), radial_coeffs=np.array([0.1, 0.01, 0.0, 0.0], np.float32), max_angle=1.3)
    cam, _, _ = camera.camera_from_batch(_batch(intrinsics_OpenCVFisheyeCameraModelParameters=fish))
    assert cam.model == abi.CAMERA_OPENCV_FISHEYE and cam.shutter == abi.SHUTTER_GLOBAL and abs(cam.max_angle - 1.3) < 1e-6
    assert np.allclose(list(cam.radial)[:4], fish["radial_coeffs"]) and list(cam.radial)[4:] == [0.0, 0.0]
    ft = dict(resolution=np.array([64, 48], np.uint32), shutter_type="ROLLING_RIGHT_TO_LEFT", principal_point=np.array([32.0, 24.0], np.float32),
              reference_poly="ANGLE_TO_PIXELDIST", pixeldist_to_angle_poly=np.array([0, 0.02, 0, 0, 0, 0], np.float32),
              angle_to_pixeldist_poly=np.array([0, 50.0, 0, 0, 0, 0], np.float32), max_angle=1.1, linear_cde=np.array([1.0, 0.0, 0.0], np.float32))
    cam, _, _ = camera.camera_from_batch(_batch(intrinsics_FThetaCameraModelParameters=ft))
    assert cam.model == abi.CAMERA_FTHETA and cam.shutter == abi.SHUTTER_ROLLING_RIGHT_TO_LEFT
    assert cam.ftheta_reference_poly == abi.FTHETA_ANGLE_TO_PIXELDIST and list(cam.ftheta_linear_cde) == [1.0, 0.0, 0.0]
    with pytest.raises(ValueError):          # tracer.py:486-488: no camera model in the batch
        camera.camera_from_batch(_batch())
    # rays already in world space: identity sensor poses (tracer.py:395-411)
    _, ps, pe = camera.camera_from_batch(_batch(intrinsics=[50.0, 50.0, 4.0, 3.0], rays_in_world_space=True))
    assert list(ps) == [0, 0, 0, 0, 0, 0, 1] and list(pe) == [0, 0, 0, 0, 0, 0, 1]


def test_configuration_defaults_and_overrides():
    """conf.render.* -> GutConfig: an empty conf gives the values of configs/render/3dgut.yaml (+ 3dgrt.yaml it inherits from);
    dict and attribute configs are equivalent; unsupported settings raise instead of silently diverging."""
    gt = importlib.import_module("3dgrut_amd.gut_tracer")
    cfg = gt.gut_config_from_conf({"render": {"splat": {}}})
    assert cfg.particle_kernel_degree == 2 and abs(cfg.particle_kernel_min_response - 0.0113) < 1e-9
    assert abs(cfg.particle_kernel_min_alpha - 1.0 / 255.0) < 1e-9 and abs(cfg.particle_kernel_max_alpha - 0.99) < 1e-7
    assert abs(cfg.min_transmittance - 1e-4) < 1e-10 and cfg.particle_radiance_sph_degree == 3
    assert (cfg.ut_alpha, cfg.ut_beta, cfg.ut_kappa) == (1.0, 2.0, 0.0) and abs(cfg.ut_in_image_margin_factor - 0.1) < 1e-8
    assert cfg.n_rolling_shutter_iterations == 5 and cfg.k_buffer_size == 0
    assert cfg.global_z_order == cfg.rect_bounding == cfg.tight_opacity_bounding == cfg.tile_based_culling == 1
    over = {"render": {"particle_kernel_degree": 4, "min_transmittance": 0.01, "splat": {"k_buffer_size": 16, "tile_based_culling": False}}}
    a = gt.gut_config_from_conf(over)
    ns = SimpleNamespace(render=SimpleNamespace(particle_kernel_degree=4, min_transmittance=0.01,
                                                splat=SimpleNamespace(k_buffer_size=16, tile_based_culling=False)))
    b = gt.gut_config_from_conf(ns)
    assert bytes(a) == bytes(b) and a.particle_kernel_degree == 4 and a.k_buffer_size == 16 and a.tile_based_culling == 0
    half = gt.gut_config_from_conf({"render": {"particle_feature_half": True, "splat": {}}})   # (setup_3dgut.py:60-61)
    assert (half.particle_feature_half, half.feature_output_half) == (1, 0) and (cfg.particle_feature_half, cfg.feature_output_half) == (0, 0)
    assert gt.fused_activations_requested({"render": {"fused_activations": True}}) and not gt.fused_activations_requested({"render": {}})
    # model.feature_type = nht -> the macro set of threedgrut/model/features.py:133-175 (defaults of configs/base_gs.yaml:96-103)
    assert cfg.feature_transform_type == 0
    nht = gt.gut_config_from_conf({"render": {"splat": {}}, "model": {"feature_type": "nht", "nht_features": {
        "dim": 48, "activation": {"type": "sincos", "num_frequencies": 1}, "interpolation_type": "barycentric"}}})
    assert (nht.feature_transform_type, nht.particle_feature_dim, nht.interp_point_feature_dim, nht.feature_interpolation_support,
            nht.feature_activation_type, nht.feature_activation_num_frequencies) == (1, 48, 12, 1, 2, 1)


def test_tracers_refuse_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    for mod, conf in (("3dgrut_amd.gut_tracer", {"render": {"splat": {}}}), ("3dgrut_amd.grt_tracer", {"render": {}})):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            importlib.import_module(mod).Tracer(conf)


def test_drop_in_package_names_resolve_to_the_plugins():
    """model.py:23,25 `import threedgrt_tracer` / `import threedgut_tracer`, each exporting `Tracer` (their __init__.py:15-17):
    with <repo>/shims first on PYTHONPATH those names bind to this repository's classes."""
    code = ("import threedgut_tracer, threedgrt_tracer; "
            "print(threedgut_tracer.Tracer.__module__, threedgrt_tracer.Tracer.__module__, threedgut_tracer.__all__, threedgrt_tracer.__all__)")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "shims"))
    out = subprocess.check_output([sys.executable, "-c", code], env=env, cwd="/tmp").decode().split()
    assert out[0] == "3dgrut_amd.gut_tracer" and out[1] == "3dgrut_amd.grt_tracer"
    assert "Tracer" in out[2] and "Tracer" in out[3]


def test_grt_configuration_defaults_and_unsupported_pipelines():
    """conf.render.* -> GrtConfig with the values of configs/render/3dgrt.yaml; pipelines / proxy primitives other than the
    reference's defaults (`reference`, `instances`) raise rather than render something else."""
    grt = importlib.import_module("3dgrut_amd.grt_tracer")
    cfg = grt.grt_config_from_conf({"render": {}})
    assert cfg.particle_kernel_degree == 4 and abs(cfg.particle_kernel_min_response - 0.0113) < 1e-9
    assert abs(cfg.particle_kernel_max_alpha - 0.99) < 1e-7 and cfg.particle_kernel_density_clamping == 1
    assert cfg.particle_radiance_sph_degree == 3 and cfg.max_hits_per_trace == 16
    assert grt.grt_config_from_conf({"render": {"particle_kernel_degree": 2}}).particle_kernel_degree == 2
    half = grt.grt_config_from_conf({"render": {"feature_output_half": True}})   # (setup_3dgrt.py:41-44)
    assert (half.particle_feature_half, half.feature_output_half) == (0, 1) and cfg.feature_output_half == 0
    # the Slang pipelines' names are accepted (same function, same kernels); other pipelines and proxies are refused
    grt.grt_config_from_conf({"render": {"pipeline_type": "referenceSlang", "backward_pipeline_type": "referenceSlangBwd"}})
    grt.grt_config_from_conf({"render": {"pipeline_type": "reference", "backward_pipeline_type": "referenceBwd"}})
    # the closed triangle-mesh proxies of particlePrimitives.cu (icosahedron = the paper's configuration), the custom primitives and the flat trisurfel proxies are provided
    assert [grt.grt_config_from_conf({"render": {"primitive_type": p}}).primitive_type
            for p in ("instances", "icosahedron", "octahedron", "tetrahedron", "diamond", "custom", "trisurfel", "trihexa", "sphere")] == [0, 1, 2, 3, 4, 5, 6, 7, 8]
    # the surfel forward pipeline (round 6): trisurfel proxies only, ten hits per trace
    bary = grt.grt_config_from_conf({"render": {"pipeline_type": "barycentricSurfels", "primitive_type": "trisurfel"}})
    assert (bary.pipeline_type, bary.primitive_type, bary.max_hits_per_trace) == (1, 6, 10) and cfg.pipeline_type == 0
    for bad in ({"pipeline_type": "fullStochastic"}, {"primitive_type": "dodecahedron"}, {"backward_pipeline_type": "referenceB2FSlangBwd"},
                {"pipeline_type": "barycentricSurfels"}, {"pipeline_type": "barycentricSurfels", "primitive_type": "icosahedron"}):
        with pytest.raises(NotImplementedError):
            grt.grt_config_from_conf({"render": bad})


def test_bench_launches_itself_for_multi_gpu_runs():
    """`python bench.py --gpus N` from a bare shell must start its own ranks (torch.distributed.run); without enough devices for RCCL it
    says so instead of dying in a launcher it never started."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GRUT_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout), r.stderr[-500:]
