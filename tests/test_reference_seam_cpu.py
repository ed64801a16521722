"""The drop-in seam exercised with the reference's OWN classes: `threedgrut.model.model.MixtureOfGaussians` and
`threedgrut.datasets.protocols.Batch` are imported from /root/reference (only where that checkout exists: this container, not the GPU
box), with `shims/` on the path so that the reference's unchanged `import threedgut_tracer` / `import threedgrt_tracer` resolve to
this repository's plugins, and the reference model constructs them from its own `conf`, calls `build_acc` and `forward` exactly as
its trainer does (model.py:262-274, 907-916).

There is no GPU here, so the layer below the plugin — the ctypes handle and the packing kernel — is replaced by a recording fake
that checks every tensor the plugin hands to the C-ABI (shapes, dtypes, contiguity, frame fields) and returns tensors of the right
shapes; everything above it (the Tracer class, its configuration parsing, camera marshalling, the autograd Function, the output
dict) is the product code.  What the kernels compute is covered by the GPU tests; this test covers that a user of the reference
can switch packages and find the objects still fit.

Third-party packages of the reference that are not installed here (omegaconf, plyfile, ncore, cv2, kornia, imageio, simplejpeg,
tensorboard; tomllib on Python 3.10) are replaced by import stubs: none of them is on the render path.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import pytest
import torch

REFERENCE = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "threedgrut")), reason="the reference checkout is only present in the build container")

_STUB_PACKAGES = ("ncore", "cv2", "imageio", "kornia", "simplejpeg", "plyfile")


class _DictConfig(dict):
    """Attribute-style access like omegaconf.DictConfig (what the reference's code expects of `conf`)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return _DictConfig(v) if isinstance(v, dict) and not isinstance(v, _DictConfig) else v

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return getattr(self, k) if k in self else default


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        return importlib.machinery.ModuleSpec(name, self, is_package=True) if name.split(".")[0] in _STUB_PACKAGES else None

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__, m.__spec__, m.__name__ = [], spec, spec.name
        return m

    def exec_module(self, module):
        pass


@pytest.fixture()
def reference(monkeypatch):
    """sys.path / sys.modules arranged like a user's environment after switching packages: reference on the path, shims shadowing the
    two tracer packages; everything is undone afterwards."""
    saved = dict(sys.modules)
    finder = _StubFinder()
    sys.meta_path.insert(0, finder)
    oc = types.ModuleType("omegaconf")

    class OmegaConf:
        create = staticmethod(lambda d=None: _DictConfig(d or {}))
        to_container = staticmethod(lambda c, resolve=True: dict(c))
        register_new_resolver = staticmethod(lambda *a, **k: None)
        is_config = staticmethod(lambda c: isinstance(c, _DictConfig))

    oc.OmegaConf, oc.DictConfig, oc.ListConfig = OmegaConf, _DictConfig, list
    oc.dictconfig = types.ModuleType("omegaconf.dictconfig")
    oc.dictconfig.DictConfig = _DictConfig
    for name, mod in (("omegaconf", oc), ("omegaconf.dictconfig", oc.dictconfig), ("torch.utils.tensorboard", MagicMock()),
                      ("torch.utils.tensorboard.writer", MagicMock())):
        monkeypatch.setitem(sys.modules, name, mod)
    if "tomllib" not in sys.modules:
        try:
            import tomllib  # noqa: F401
        except ModuleNotFoundError:
            monkeypatch.setitem(sys.modules, "tomllib", importlib.import_module("tomli"))
    for p in (REFERENCE, ROOT, os.path.join(ROOT, "shims")):
        monkeypatch.syspath_prepend(p)
    yield
    sys.meta_path.remove(finder)
    for k in list(sys.modules):
        if k not in saved and (k.split(".")[0] in ("threedgrut", "threedgut_tracer", "threedgrt_tracer", "omegaconf") + _STUB_PACKAGES):
            del sys.modules[k]


_REAL = {}   # the product's native wrapper classes, captured before they are replaced by the recorders


class _Recorder:
    """Stands in for the C handle: checks what crosses the boundary, returns tensors of the contract's shapes."""

    def __init__(self, cfg, kind):
        self.cfg, self.kind, self.calls = cfg, kind, []
        self.ncoef = (cfg.particle_radiance_sph_degree + 1) ** 2

    def _check_particles(self, pd, sph, n):
        assert pd.shape == (n, 12) and pd.dtype == torch.float32 and pd.is_contiguous()
        assert sph.shape == (n, 3 * self.ncoef) and sph.dtype == torch.float32 and sph.is_contiguous()

    def collect_times(self):
        return {}


class _GutRecorder(_Recorder):
    def __init__(self, cfg):
        super().__init__(cfg, "gut")
        self.make_frame = lambda *a: _REAL["gut"].make_frame(self, *a)   # the product's own marshalling (pure ctypes)

    def trace(self, frame, pd, sph, ro, rd):
        H, W, n = frame.height, frame.width, frame.num_particles
        self._check_particles(pd, sph, n)
        assert ro.shape == (1, H, W, 3) and rd.shape == (1, H, W, 3) and ro.is_contiguous() and rd.dtype == torch.float32
        assert frame.camera.width == W and frame.camera.height == H and frame.n_active_features >= 0
        self.calls.append(("trace", frame.frame_id))
        fd = torch.rand((H, W, 4))
        return fd, torch.rand((H, W, 1)), torch.zeros((H, W, 1)), torch.ones((n, 1)), fd[..., :3].contiguous(), fd[..., 3:].contiguous()

    def trace_bwd_unpacked(self, frame, pd, sph, ro, rd, fd, g_feat, g_opa, dist, g_dist):
        n = frame.num_particles
        assert g_feat is None or (g_feat.shape == fd.shape[:2] + (3,) and g_feat.is_contiguous())
        assert g_opa is None or g_opa.shape == fd.shape[:2] + (1,)
        self.calls.append(("trace_bwd", g_dist is not None))
        return torch.ones((n, 3)), torch.ones((n, 1)), torch.ones((n, 4)), torch.ones((n, 3)), torch.ones_like(sph)


class _GrtRecorder(_Recorder):
    def __init__(self, cfg):
        super().__init__(cfg, "grt")
        self.make_frame = lambda *a: _REAL["grt"].make_frame(self, *a)

    def build_bvh(self, pos, rot, scl, dns, rebuild, allow_update):
        n = pos.shape[0]
        assert pos.shape == (n, 3) and rot.shape == (n, 4) and scl.shape == (n, 3) and dns.shape == (n, 1)
        assert torch.allclose(rot.norm(dim=1), torch.ones(n), atol=1e-5) and bool((scl > 0).all()) and bool(((dns > 0) & (dns < 1)).all())
        self.calls.append(("build_bvh", bool(rebuild)))

    def trace(self, frame, pd, sph, ro, rd, hit_capacity=0):
        H, W, n = frame.height, frame.width, frame.num_particles
        self._check_particles(pd, sph, n)
        assert len(list(frame.ray_to_world)) == 12 and frame.min_transmittance > 0
        self.calls.append(("trace", frame.keep_hits_for_backward))
        return (torch.rand((1, H, W, 3)), torch.rand((1, H, W, 1)), torch.rand((1, H, W, 2)), torch.rand((1, H, W, 3)), torch.zeros((1, H, W, 1)),
                torch.ones((n, 1)))

    def trace_bwd(self, frame, pd, sph, ro, rd, feat, dns, hit, nrm, g_feat, g_dns, g_hit, g_nrm):
        assert g_feat.shape == feat.shape and g_dns.shape == dns.shape
        self.calls.append(("trace_bwd", g_hit is not None))
        return torch.ones_like(pd), torch.ones_like(sph)


def _conf(method):
    return _DictConfig({
        "model": {"feature_type": "sh", "density_activation": "sigmoid", "scale_activation": "exp",
                  "background": {"name": "skip-background", "color": "black"},
                  "progressive_training": {"init_n_features": 1, "increase_frequency": 1000, "increase_step": 1, "max_n_features": 3}},
        "render": {"method": method, "particle_radiance_sph_degree": 3, "particle_kernel_degree": 2 if method == "3dgut" else 4,
                   "particle_kernel_density_clamping": True, "min_transmittance": 1e-4 if method == "3dgut" else 1e-3, "enable_normals": False,
                   "enable_hitcounts": True, "enable_kernel_timings": False, "pipeline_type": "reference", "primitive_type": "instances",
                   "max_consecutive_bvh_update": 15, "splat": {"k_buffer_size": 0, "global_z_order": True}},
    })


@pytest.mark.parametrize("method", ["3dgut", "3dgrt"])
def test_reference_model_and_batch_drive_the_plugin_unchanged(reference, monkeypatch, method):
    abi = importlib.import_module("3dgrut_amd._abi")
    gt = importlib.import_module("3dgrut_amd.gut_tracer")
    grt = importlib.import_module("3dgrut_amd.grt_tracer")
    _REAL.update(gut=gt._GutNative, grt=grt._GrtNative)
    monkeypatch.setattr(gt, "_GutNative", _GutRecorder)
    monkeypatch.setattr(grt, "_GrtNative", _GrtRecorder)

    def pack(pos, dns, rot, scl):   # grut_pack_particles on the host: [N,12] rows {pos, density, quat wxyz, scale, 0}
        return torch.cat([pos, dns, rot, scl, torch.zeros_like(dns)], dim=1)

    monkeypatch.setattr(abi, "pack_particles", pack)
    monkeypatch.setattr(abi, "unpack_particle_grads", lambda g: (g[:, 0:3].contiguous(), g[:, 3:4].contiguous(), g[:, 4:8].contiguous(), g[:, 8:11].contiguous()))

    model_mod = importlib.import_module("threedgrut.model.model")          # the reference's module, its own imports untouched
    Batch = importlib.import_module("threedgrut.datasets.protocols").Batch
    assert importlib.import_module("threedgut_tracer").Tracer is gt.Tracer and importlib.import_module("threedgrt_tracer").Tracer is grt.Tracer

    mog = model_mod.MixtureOfGaussians(_conf(method), scene_extent=1.0)    # constructs OUR Tracer through `import threed*_tracer`
    assert type(mog.renderer) is (gt.Tracer if method == "3dgut" else grt.Tracer)
    n, H, W = 50, 20, 28
    g = torch.Generator().manual_seed(0)
    P = torch.nn.Parameter
    mog.positions, mog.rotation = P(torch.randn((n, 3), generator=g)), P(torch.randn((n, 4), generator=g))
    mog.scale, mog.density = P(torch.randn((n, 3), generator=g) - 3), P(torch.randn((n, 1), generator=g))
    mog.features_albedo, mog.features_specular = P(torch.rand((n, 3), generator=g)), P(torch.zeros((n, 45)))
    mog.build_acc(rebuild=True)                                              # model.py:272-274
    d = torch.nn.functional.normalize(torch.randn((1, H, W, 3), generator=g), dim=-1)
    batch = Batch(rays_ori=torch.zeros((1, H, W, 3)), rays_dir=d, T_to_world=torch.eye(4)[None], intrinsics=[30.0, 30.0, W / 2, H / 2])
    out = mog(batch, train=True, frame_id=3)                                  # model.py:907-916 -> Tracer.render
    assert set(out) >= {"pred_features", "pred_opacity", "pred_dist", "pred_normals", "hits_count", "frame_time_ms", "mog_visibility"}
    assert out["pred_features"].shape == (1, H, W, 3) and out["pred_opacity"].shape == (1, H, W, 1) and out["pred_dist"].shape == (1, H, W, 1)
    assert out["pred_normals"].shape == (1, H, W, 3) and out["hits_count"].shape == (1, H, W, 1) and out["mog_visibility"].shape == (n, 1)
    assert out["pred_features"].is_contiguous() and out["pred_opacity"].is_contiguous()
    (out["pred_features"].sum() + out["pred_opacity"].sum()).backward()      # the trainer's loss touches colour and opacity only
    for name in ("positions", "rotation", "scale", "density", "features_albedo", "features_specular"):
        grad = getattr(mog, name).grad
        assert grad is not None and grad.shape == getattr(mog, name).shape and bool(torch.isfinite(grad).all()), name
    calls = [c[0] for c in mog.renderer.tracer_wrapper.calls]
    assert calls == (["trace", "trace_bwd"] if method == "3dgut" else ["build_bvh", "trace", "trace_bwd"])
    # inference through the reference's convenience entry point (model.py:918-929): world-space rays, no pose
    with torch.no_grad():
        out2 = mog.trace(torch.zeros((1, H, W, 3)), d) if method == "3dgrt" else None
    assert out2 is None or out2["pred_features"].shape == (1, H, W, 3)
