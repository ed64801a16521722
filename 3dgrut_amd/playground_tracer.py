"""Hybrid mesh + Gaussian tracer: drop-in for the render path of `threedgrut_playground.tracer.Tracer`
(threedgrut_playground/tracer.py:47-268) — `build_gs_acc`, `build_mesh_acc`, `render`, `render_playground` — over the C-ABI's
grt_build_mesh_bvh / grt_trace_hybrid (include/grut_amd.h).  BASELINE config 5: primary rays + reflection / refraction through
triangle meshes with Gaussian segments in between, forward only like the reference.

Not implemented (raises): PBR primitives (primitive type 4: Cook-Torrance sampling, glTF textures, emissive maps), the OptiX
denoiser; the environment map is reduced to its mean colour.  `materials[i].diffuseFactor` gives the base colour of diffuse faces
(PGRNDRenderDisablePBRTextures semantics)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi
from .grt_tracer import Tracer as _GrtTracer
from .gut_tracer import _ptr, _stream_ptr

PGRND_PRIMITIVE_PBR = 4


class Tracer(_GrtTracer):
    def build_gs_acc(self, gaussians, rebuild=True):
        return self.build_acc(gaussians, rebuild)

    def build_mesh_acc(self, mesh_vertices, mesh_faces, rebuild=True, allow_update=True):
        v = mesh_vertices.detach().view(-1, 3).contiguous().float()
        f = mesh_faces.detach().view(-1, 3).contiguous().to(torch.int32)
        self._mesh_vertices, self._mesh_faces = v, f   # the trace reads them again (vertex positions for hard normals)
        nat = self.tracer_wrapper
        _abi.check(nat.lib.grt_build_mesh_bvh(nat.handle, _stream_ptr(v.device), v.shape[0], _ptr(v), f.shape[0], _ptr(f)), "grt_build_mesh_bvh")

    def render_playground(self, gaussians, ray_o, ray_d, playground_opts, mesh_faces, vertex_normals, vertex_tangents, vertex_tangents_mask,
                          primitive_type, frame_id=0, ray_max_t=None, material_uv=None, material_id=None, materials=None, is_sync_materials=True,
                          refractive_index=None, envmap=None, envmap_offset=None, max_pbr_bounces=7):
        nat = self.tracer_wrapper
        dev = ray_o.device
        H, W = int(ray_o.shape[1]), int(ray_o.shape[2])
        faces = mesh_faces.detach().view(-1, 3).contiguous().to(torch.int32)
        F = faces.shape[0]
        prim = primitive_type.detach().view(-1).contiguous().to(torch.int32)
        if F and bool((prim == PGRND_PRIMITIVE_PBR).any()):
            raise NotImplementedError("3dgrut_amd: PBR primitives (Cook-Torrance sampling, textures) are not implemented in the hybrid tracer")
        refr = torch.ones(F, device=dev) if refractive_index is None else refractive_index.detach().view(-1).contiguous().float()
        # base colour of each face: its material's diffuseFactor (materials: objects with .diffuseFactor, material_id [F,1] / [F])
        diffuse = torch.full((max(F, 1), 3), 0.8, device=dev)
        if materials and material_id is not None and material_id.numel():
            table = torch.stack([torch.as_tensor(getattr(m, "diffuseFactor", getattr(m, "diffuse_factor", [0.8, 0.8, 0.8])), dtype=torch.float32)[:3]
                                 for m in materials]).to(dev)
            diffuse = table[material_id.detach().view(F, -1)[:, 0].long()].contiguous()
        background = torch.zeros(3) if envmap is None else envmap.detach().float().reshape(-1, envmap.shape[-1])[:, :3].mean(0).cpu()
        vn = None if vertex_normals is None else vertex_normals.detach().view(-1, 3).contiguous().float()
        features = gaussians.get_features()
        particle_density = _abi.pack_particles(gaussians.positions.contiguous(), gaussians.get_density().contiguous(),
                                               gaussians.get_rotation().contiguous(), gaussians.get_scale().contiguous())
        frame = nat.make_frame(frame_id, gaussians.n_active_features, self._min_transmittance, gaussians.num_gaussians, H, W,
                               torch.eye(4)[None])   # rays arrive in world coordinates (tracer.py:203)
        mesh = _abi.GrtMesh(int(self._mesh_vertices.shape[0]), F, _ptr(self._mesh_vertices), _ptr(faces), _ptr(vn), _ptr(prim), _ptr(refr), _ptr(diffuse))
        opts = _abi.GrtHybridOptions(int(playground_opts), int(max_pbr_bounces), (C.c_float * 3)(*[float(x) for x in background]))
        rgb = torch.empty((1, H, W, 3), device=dev)
        opa = torch.empty((1, H, W, 1), device=dev)
        last = torch.empty((1, H, W, 6), device=dev)
        bounces = torch.empty((1, H, W, 1), dtype=torch.int32, device=dev)
        ro, rd = ray_o.detach().contiguous().float(), ray_d.detach().contiguous().float()
        tmax = None if ray_max_t is None else ray_max_t.detach().contiguous().float()
        _abi.check(nat.lib.grt_trace_hybrid(nat.handle, _stream_ptr(dev), C.byref(frame), _ptr(particle_density), _ptr(features.contiguous()), _ptr(ro),
                                            _ptr(rd), _ptr(tmax), C.byref(mesh), C.byref(opts), _ptr(rgb), _ptr(opa), _ptr(last), _ptr(bounces)),
                   "grt_trace_hybrid")
        return {"pred_features": rgb, "pred_opacity": opa, "last_ray_o": last[..., :3].contiguous(), "last_ray_d": last[..., 3:].contiguous(),
                "mirror_bounces": bounces}
