"""Seeded scenes for the hybrid mesh + Gaussian path tracer (BASELINE config 5; threedgrut_playground): Gaussians, a triangle mesh with the
playground's per-face / per-vertex attributes, a material table with textures, an environment map, world-space rays.  Shared by
tests/golden/make_golden.py (reference programs -> tests/golden/playground.npz), the oracle tests and the GPU tests."""
import numpy as np

from .scenes import make_scene

F32 = np.float32
PRIM_NONE, PRIM_MIRROR, PRIM_GLASS, PRIM_DIFFUSE, PRIM_PBR = 0, 1, 2, 3, 4
OPT_SMOOTH_NORMALS, OPT_NO_GAUSSIANS, OPT_NO_TEXTURES = 1, 2, 4
ALPHA_OPAQUE, ALPHA_BLEND, ALPHA_MASK = 0, 1, 2


def uv_sphere(centre, radius, subdiv):
    """(vertices, faces, normals, uv per vertex)"""
    V, Fc, UV = [], [], []
    for i in range(subdiv + 1):
        th = np.pi * i / subdiv
        for j in range(2 * subdiv):
            ph = np.pi * j / subdiv
            V.append([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
            UV.append([j / (2 * subdiv), i / subdiv])
    for i in range(subdiv):
        for j in range(2 * subdiv):
            a, b = i * 2 * subdiv + j, i * 2 * subdiv + (j + 1) % (2 * subdiv)
            c, d = a + 2 * subdiv, b + 2 * subdiv
            if i > 0:
                Fc.append([a, c, b])
            if i < subdiv - 1:
                Fc.append([b, c, d])
    n = np.asarray(V, F32)
    return (n * radius + np.asarray(centre, F32)).astype(F32), np.asarray(Fc, np.int32), n, np.asarray(UV, F32)


def quad(corners, normal):
    v = np.asarray(corners, F32)
    return v, np.array([[0, 1, 2], [0, 2, 3]], np.int32), np.tile(np.asarray(normal, F32), (4, 1)), np.array([[0, 0], [1, 0], [1, 1], [0, 1]], F32)


def texture(h, w, c, seed, lo=0.0, hi=1.0):
    r = np.random.default_rng(seed)
    base = r.uniform(lo, hi, (h, w, c))
    yy, xx = np.mgrid[0:h, 0:w]
    base[..., 0] = lo + (hi - lo) * (0.5 + 0.5 * np.sin(0.9 * xx + 0.5 * yy))   # smooth structure + noise: filtering matters
    return np.ascontiguousarray(base.astype(F32))


def material(diffuse=(0.8, 0.8, 0.8, 1.0), emissive=(0, 0, 0), metallic=0.0, roughness=0.5, transmission=0.0, ior=1.5, alpha_mode=ALPHA_OPAQUE,
             alpha_cutoff=0.5, diffuse_tex=None, emissive_tex=None, metallic_roughness_tex=None, normal_tex=None):
    return dict(diffuse_factor=np.asarray(diffuse, F32), emissive_factor=np.asarray(emissive, F32), metallic_factor=float(metallic),
                roughness_factor=float(roughness), transmission_factor=float(transmission), ior=float(ior), alpha_mode=int(alpha_mode),
                alpha_cutoff=float(alpha_cutoff), diffuse_tex=diffuse_tex, emissive_tex=emissive_tex, metallic_roughness_tex=metallic_roughness_tex,
                normal_tex=normal_tex)


def _assemble(parts):
    """parts: list of (vertices, faces, normals, uv, prim_type, material_id, ior, tangents or None)"""
    V, Fc, N, T, HT, P, UV, MID, IOR = [], [], [], [], [], [], [], [], []
    base = 0
    for v, f, n, uv, prim, mid, ior, tan in parts:
        V.append(v); N.append(n)
        Fc.append(f + base)
        T.append(np.zeros_like(v) if tan is None else np.asarray(tan, F32))
        HT.append(np.full(len(v), 0 if tan is None else 1, np.uint8))
        P.append(np.full(len(f), prim, np.int32)); MID.append(np.full(len(f), mid, np.int32)); IOR.append(np.full(len(f), ior, F32))
        UV.append(uv[f])           # [F,3,2]: per face and corner
        base += len(v)
    return dict(vertices=np.ascontiguousarray(np.concatenate(V), F32), triangles=np.ascontiguousarray(np.concatenate(Fc), np.int32),
                vertex_normals=np.ascontiguousarray(np.concatenate(N), F32), vertex_tangents=np.ascontiguousarray(np.concatenate(T), F32),
                vertex_has_tangents=np.ascontiguousarray(np.concatenate(HT), np.uint8), prim_type=np.ascontiguousarray(np.concatenate(P)),
                mat_uv=np.ascontiguousarray(np.concatenate(UV), F32), mat_id=np.ascontiguousarray(np.concatenate(MID)),
                refractive_index=np.ascontiguousarray(np.concatenate(IOR)))


def make_playground_scene(kind, width=48, height=32, n=1500, seed=7):
    """kind: 'classic' (mirror sphere, glass pane, textured diffuse floor), 'pbr' (metal / dielectric-transmissive / fully textured /
    alpha-masked / alpha-blended PBR spheres over a PBR floor), 'mixed' (both families in one frame)."""
    sc = make_scene(n=n, width=width, height=height, median_scale=0.07, max_density=0.6, seed=seed)
    T = np.asarray(sc["batch"]["T_to_world"][0], F32)
    ro, rd = (np.asarray(a, F32).reshape(height, width, 3) for a in sc["rays"])
    ro_w = (ro @ T[:3, :3].T + T[:3, 3]).astype(F32)   # the playground hands over world-space rays (tracer.py:203)
    rd_w = (rd @ T[:3, :3].T).astype(F32)
    centre = (T[:3, 3] + 2.2 * (T[:3, :3] @ np.array([0, 0, 1.0]))).astype(F32)   # in front of the camera, inside the cloud's view
    right, up, fwd = T[:3, 0], T[:3, 1], T[:3, 2]
    parts, mats = [], []
    if kind in ("classic", "mixed"):
        mats.append(material(diffuse=(0.9, 0.8, 0.7, 1.0), diffuse_tex=texture(8, 8, 4, seed + 1, 0.2, 1.0)))       # 0: the diffuse floor
        v, f, nn, uv = uv_sphere(centre - 0.45 * right, 0.32, 7)
        parts.append((v, f, nn, uv, PRIM_MIRROR, 0, 1.0, None))
        c = centre + 0.45 * right - 0.5 * fwd
        gv = [c - 0.3 * right - 0.3 * up, c + 0.3 * right - 0.3 * up + 0.1 * fwd, c + 0.3 * right + 0.3 * up + 0.1 * fwd, c - 0.3 * right + 0.3 * up]
        v, f, nn, uv = quad(gv, -fwd)
        parts.append((v, f, nn, uv, PRIM_GLASS, 0, 1.45, None))
        c = centre + 0.75 * up + 0.3 * fwd
        fv = [c - 1.5 * right - 0.2 * fwd, c + 1.5 * right - 0.2 * fwd, c + 1.5 * right + 1.2 * fwd + 0.3 * up, c - 1.5 * right + 1.2 * fwd + 0.3 * up]
        v, f, nn, uv = quad(fv, -up)
        parts.append((v, f, nn, uv, PRIM_DIFFUSE, 0, 1.0, None))
    if kind in ("pbr", "mixed"):
        m0 = len(mats)
        mats += [material(diffuse=(0.95, 0.75, 0.3, 1.0), metallic=1.0, roughness=0.25),                                                # gold-like metal
                 material(diffuse=(0.9, 0.95, 1.0, 1.0), metallic=0.0, roughness=0.1, transmission=0.85, ior=1.4),                      # transmissive dielectric
                 material(diffuse=(1.0, 1.0, 1.0, 1.0), emissive=(0.6, 0.5, 0.4), metallic=0.8, roughness=0.9,
                          diffuse_tex=texture(8, 16, 4, seed + 2, 0.1, 1.0), emissive_tex=texture(4, 4, 4, seed + 3, 0.0, 0.5),
                          metallic_roughness_tex=texture(8, 8, 2, seed + 4, 0.05, 1.0), normal_tex=texture(8, 8, 4, seed + 5, 0.0, 1.0)),   # everything textured
                 material(diffuse=(0.3, 0.8, 0.4, 0.45), roughness=0.6, alpha_mode=ALPHA_MASK, alpha_cutoff=0.5),                        # masked out: "no material"
                 material(diffuse=(0.8, 0.3, 0.3, 0.5), roughness=0.4, alpha_mode=ALPHA_BLEND),                                          # stochastic alpha
                 material(diffuse=(0.6, 0.6, 0.65, 1.0), roughness=0.8, diffuse_tex=texture(16, 16, 4, seed + 6, 0.3, 0.9))]            # floor
        offs = [(-0.6, -0.25), (-0.1, 0.2), (0.45, -0.2), (0.15, -0.45), (-0.3, 0.5)] if kind == "pbr" else [(0.0, -0.55), (0.3, 0.45), (-0.1, 0.1)]
        for k, (ox, oy) in enumerate(offs):
            v, f, nn, uv = uv_sphere(centre + ox * right + oy * up + (0.15 * k - 0.2) * fwd, 0.2, 6)
            tan = None
            if k == 2:   # precomputed vertex tangents on one object (materials.cuh:104-135)
                tan = np.cross(np.array([0.0, 0.0, 1.0]), nn)
                tan[np.linalg.norm(tan, axis=1) < 1e-6] = [1.0, 0.0, 0.0]
                tan = tan / np.linalg.norm(tan, axis=1, keepdims=True)
            parts.append((v, f, nn, uv, PRIM_PBR, m0 + k, 1.0, tan))
        c = centre + 0.8 * up + 0.3 * fwd
        fv = [c - 1.5 * right - 0.3 * fwd, c + 1.5 * right - 0.3 * fwd, c + 1.5 * right + 1.2 * fwd, c - 1.5 * right + 1.2 * fwd]
        v, f, nn, uv = quad(fv, -up)
        parts.append((v, f, nn, uv, PRIM_PBR, m0 + 5, 1.0, None))
    mesh = _assemble(parts)
    env = texture(8, 16, 4, seed + 9, 0.05, 0.9)
    return dict(density12=np.ascontiguousarray(sc["density12"], F32), sph=np.ascontiguousarray(sc["sph"], F32), ray_o=np.ascontiguousarray(ro_w),
                ray_d=np.ascontiguousarray(rd_w), W=width, H=height, mesh=mesh, materials=mats, envmap=env, envmap_offset=np.array([0.13, 0.04], F32),
                ray_max_t=np.full((height, width), 1e9, F32))


# the configurations stored in tests/golden/playground.npz: (name, scene kind, playground_opts, max_pbr_bounces, frame_number)
GOLDEN_CASES = [("classic_smooth", "classic", OPT_SMOOTH_NORMALS, 7, 0),
                ("classic_hard_notex", "classic", OPT_NO_TEXTURES, 7, 3),
                ("pbr_smooth", "pbr", OPT_SMOOTH_NORMALS, 4, 1),
                ("pbr_hard_nogauss", "pbr", OPT_NO_GAUSSIANS, 6, 5),
                ("mixed_smooth", "mixed", OPT_SMOOTH_NORMALS, 5, 2)]
