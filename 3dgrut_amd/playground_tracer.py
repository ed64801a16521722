"""Hybrid mesh + Gaussian tracer: drop-in for the render path of `threedgrut_playground.tracer.Tracer`
(threedgrut_playground/tracer.py:47-268) — `build_gs_acc`, `build_mesh_acc`, `render`, `render_playground` — over the C-ABI's
grt_build_mesh_bvh / grt_trace_hybrid (include/grut_amd.h).  BASELINE config 5: primary rays + reflection / refraction / PBR scattering
through triangle meshes with Gaussian segments in between, forward only like the reference.

Materials are the reference's `Material` objects (threedgrut_playground/utils/mesh_io.py; attribute names as tracer.py:133-170 reads
them: diffuse_map, emissive_map, metallic_roughness_map, normal_map, diffuse_factor, emissive_factor, metallic_factor,
roughness_factor, alpha_mode, alpha_cutoff, transmission_factor, ior) or plain dicts with those keys.  Not provided: the OptiX
denoiser (`denoise`)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi
from .grt_tracer import Tracer as _GrtTracer, features_for_kernel
from .gut_tracer import _ptr, _stream_ptr


def _mat_get(m, key, default=None):
    v = m.get(key, None) if isinstance(m, dict) else getattr(m, key, None)
    return default if v is None else v


def _texture(t, channels, device, keep):
    """[H,W,C] tensor -> GrtTexture on `device` (an empty / missing map = no texture, hybridTracer.cpp:250-286)."""
    if t is None or not torch.is_tensor(t) or t.numel() == 0 or t.dim() < 2 or t.shape[0] == 0 or t.shape[1] == 0:
        return _abi.GrtTexture(None, 0, 0, channels)
    t = t.detach().to(device=device, dtype=torch.float32).reshape(t.shape[0], t.shape[1], -1)
    if t.shape[2] < channels:   # (tracer.py:136-145 pads missing channels with ones)
        t = torch.cat([t, torch.ones(t.shape[0], t.shape[1], channels - t.shape[2], device=device)], dim=2)
    t = t[..., :channels].contiguous()
    keep.append(t)
    return _abi.GrtTexture(t.data_ptr(), int(t.shape[0]), int(t.shape[1]), channels)


def native_materials(materials, device, keep):
    """The reference's to_native_pbr_material (tracer.py:133-170) -> a ctypes array of GrtMaterial (defaults as there)."""
    n = len(materials)
    arr = (_abi.GrtMaterial * max(n, 1))()
    for i, m in enumerate(materials):
        o = arr[i]
        o.diffuse = _texture(_mat_get(m, "diffuse_map"), 4, device, keep)
        o.emissive = _texture(_mat_get(m, "emissive_map"), 4, device, keep)
        o.metallic_roughness = _texture(_mat_get(m, "metallic_roughness_map"), 2, device, keep)
        o.normal = _texture(_mat_get(m, "normal_map"), 4, device, keep)
        df = [float(x) for x in torch.as_tensor(_mat_get(m, "diffuse_factor", [1.0, 1.0, 1.0, 1.0])).flatten().tolist()]
        df = (df + [1.0] * 4)[:4]
        ef = [float(x) for x in torch.as_tensor(_mat_get(m, "emissive_factor", [0.0, 0.0, 0.0])).flatten().tolist()]
        ef = (ef + [0.0] * 3)[:3]
        for k in range(4):
            o.diffuse_factor[k] = df[k]
        for k in range(3):
            o.emissive_factor[k] = ef[k]
        o.metallic_factor = float(_mat_get(m, "metallic_factor", 0.0) or 0.0)
        o.roughness_factor = float(_mat_get(m, "roughness_factor", 0.0) or 0.0)
        o.alpha_mode = int(_mat_get(m, "alpha_mode", 0) or 0)
        o.alpha_cutoff = float(_mat_get(m, "alpha_cutoff", 0.5) or 0.5)
        o.transmission_factor = float(_mat_get(m, "transmission_factor", 0.0) or 0.0)
        o.ior = float(_mat_get(m, "ior", 0.0) or 0.0)
    return arr, n


class Tracer(_GrtTracer):
    def build_gs_acc(self, gaussians, rebuild=True):
        return self.build_acc(gaussians, rebuild)

    def build_mesh_acc(self, mesh_vertices, mesh_faces, rebuild=True, allow_update=True):
        v = mesh_vertices.detach().view(-1, 3).contiguous().float()
        f = mesh_faces.detach().view(-1, 3).contiguous().to(torch.int32)
        self._mesh_vertices, self._mesh_faces = v, f   # the trace reads them again (vertex positions for hard normals)
        nat = self.tracer_wrapper
        _abi.check(nat.lib.grt_build_mesh_bvh(nat.handle, _stream_ptr(v.device), v.shape[0], _ptr(v), f.shape[0], _ptr(f), int(bool(rebuild)),
                                              int(bool(allow_update))), "grt_build_mesh_bvh")

    def render_playground(self, gaussians, ray_o, ray_d, playground_opts, mesh_faces, vertex_normals, vertex_tangents, vertex_tangents_mask,
                          primitive_type, frame_id=0, ray_max_t=None, material_uv=None, material_id=None, materials=None, is_sync_materials=True,
                          refractive_index=None, envmap=None, envmap_offset=None, max_pbr_bounces=7):
        nat = self.tracer_wrapper
        dev = ray_o.device
        H, W = int(ray_o.shape[1]), int(ray_o.shape[2])
        faces = mesh_faces.detach().view(-1, 3).contiguous().to(torch.int32)
        F = faces.shape[0]
        V = int(self._mesh_vertices.shape[0])
        keep = []
        prim = primitive_type.detach().view(-1).contiguous().to(device=dev, dtype=torch.int32)
        refr = torch.ones(F, device=dev) if refractive_index is None else refractive_index.detach().view(-1).contiguous().to(device=dev, dtype=torch.float32)
        vn = None if vertex_normals is None else vertex_normals.detach().view(-1, 3).contiguous().to(device=dev, dtype=torch.float32)
        vt = vh = None
        if vertex_tangents is not None and vertex_tangents_mask is not None and vertex_tangents.numel() == 3 * V:
            vt = vertex_tangents.detach().view(-1, 3).contiguous().to(device=dev, dtype=torch.float32)
            vh = vertex_tangents_mask.detach().view(-1).contiguous().to(device=dev, dtype=torch.uint8)
        uv = None
        if material_uv is not None and material_uv.numel() == 6 * F:
            uv = material_uv.detach().view(-1, 3, 2).contiguous().to(device=dev, dtype=torch.float32)
        mid = None
        if material_id is not None and material_id.numel() >= F and F:
            mid = material_id.detach().view(F, -1)[:, 0].contiguous().to(device=dev, dtype=torch.int32)
        mats, n_mats = native_materials(materials or [], dev, keep)
        env = _texture(envmap, 4, dev, keep) if (envmap is not None and envmap.dim() == 3) else _abi.GrtTexture(None, 0, 0, 4)
        off = [0.0, 0.0] if envmap_offset is None else [float(x) for x in envmap_offset.detach().flatten().tolist()[:2]]
        sph = features_for_kernel(nat.cfg, gaussians.get_features())
        particle_density = _abi.pack_particles(gaussians.positions.contiguous(), gaussians.get_density().contiguous(),
                                               gaussians.get_rotation().contiguous(), gaussians.get_scale().contiguous())
        frame = nat.make_frame(frame_id, gaussians.n_active_features, self._min_transmittance, gaussians.num_gaussians, H, W,
                               torch.eye(4)[None])   # rays arrive in world coordinates (tracer.py:203)
        mesh = _abi.GrtMesh(V, F, _ptr(self._mesh_vertices), _ptr(faces), _ptr(vn), _ptr(vt), _ptr(vh), _ptr(prim), _ptr(uv), _ptr(mid), _ptr(refr),
                            n_mats, mats, env, (C.c_float * 2)(*off))
        opts = _abi.GrtHybridOptions(int(playground_opts), int(max_pbr_bounces), int(frame_id) & 0xFFFFFFFF)
        rgb = torch.empty((1, H, W, 3), device=dev)
        opa = torch.empty((1, H, W, 1), device=dev)
        last = torch.empty((1, H, W, 6), device=dev)
        bounces = torch.empty((1, H, W, 1), dtype=torch.int32, device=dev)
        ro, rd = ray_o.detach().contiguous().float(), ray_d.detach().contiguous().float()
        tmax = None if ray_max_t is None else ray_max_t.detach().contiguous().float()
        _abi.check(nat.lib.grt_trace_hybrid(nat.handle, _stream_ptr(dev), C.byref(frame), _ptr(particle_density), _ptr(sph), _ptr(ro),
                                            _ptr(rd), _ptr(tmax), C.byref(mesh), C.byref(opts), _ptr(rgb), _ptr(opa), _ptr(last), _ptr(bounces)),
                   "grt_trace_hybrid")
        zeros = lambda c: torch.zeros((1, H, W, c), device=dev)
        # (the reference's raygen writes radiance, density and the last ray only: its hit distance / normal / hit count outputs stay as
        # allocated, zeros — hybridTracer.cpp traceHybrid, trace.cuh:136-173)
        return {"pred_features": rgb, "pred_opacity": opa, "pred_dist": zeros(1), "pred_normals": zeros(3), "hits_count": zeros(1),
                "last_ray_o": last[..., :3].contiguous(), "last_ray_d": last[..., 3:].contiguous(), "mirror_bounces": bounces}
