"""Golden gradients for the two backward stages of 3DGUT that exist in the reference only as Slang autodiff OUTPUT (the generated
header is not in the checkout): the sorted K > 0 compositing backward and the projection backward (G12).

    python tests/golden/make_autograd_golden.py        ->  tests/golden/autograd_gut.npz

Nothing here is derived by hand: the reference FORWARD is restated in float64 torch from its Slang / CUDA sources and
torch.autograd differentiates it, exactly what slangc's reverse mode does to the same forward at the reference's build time:

  * sphericalHarmonics.decode          threedgut_tracer/include/3dgut/kernels/slang/common/sphericalHarmonics.slang:21-64
  * per-particle incident direction    threedgut_tracer/include/3dgut/kernels/cuda/renderers/gutProjector.cuh:304-310
  * gaussianParticle.hit               .../slang/models/gaussianParticles.slang:96-110 (canonical ray), :112-168 (kernel response),
                                       :181-190 (hit distance), :207-242 (accept test, alpha clamp)
  * integrateHit / integrateRadiance   .../slang/models/gaussianParticles.slang:244-274, shRadiativeParticles.slang:83-99
  * hit k-buffer + ray loop            .../cuda/renderers/gutKBufferRenderer.cuh:62-122 (insertion), :273-352 (evalKBuffer)
  * quaternion -> rotation^T           .../slang/common/transforms.slang:22-39

The per-tile particle lists (which particle is offered to which pixel, in which order) are integer data and come from the oracle's
binning, which tests/golden/projector.npz pins to the reference's own projector code bit for bit.  The forward VALUE computed here
is checked against the oracle's float64 forward before anything is written (must agree to 1e-10).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from scenes import make_scene  # noqa: E402

torch.set_default_dtype(torch.float64)
MIN_RESPONSE, MIN_ALPHA, MAX_ALPHA, MIN_T = 0.0113, 1.0 / 255.0, 0.99, 1e-4   # configs/render/3dgut.yaml, threedgut.cuh

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
      -0.5900435899266435]


def sh_decode(coeffs, d):
    """sphericalHarmonics.decode, degree 3.  coeffs [N,16,3], d [N,3] -> [N,3] (clamped at 0 after the +0.5 shift)."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    c = coeffs
    f = C0 * c[:, 0]
    f = f - C1 * y * c[:, 1] + C1 * z * c[:, 2] - C1 * x * c[:, 3]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    f = f + C2[0] * xy * c[:, 4] + C2[1] * yz * c[:, 5] + C2[2] * (2.0 * zz - xx - yy) * c[:, 6] + C2[3] * xz * c[:, 7] + C2[4] * (xx - yy) * c[:, 8]
    f = (f + C3[0] * y * (3.0 * xx - yy) * c[:, 9] + C3[1] * xy * z * c[:, 10] + C3[2] * y * (4.0 * zz - xx - yy) * c[:, 11]
         + C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * c[:, 12] + C3[4] * x * (4.0 * zz - xx - yy) * c[:, 13] + C3[5] * z * (xx - yy) * c[:, 14]
         + C3[6] * x * (xx - 3.0 * yy) * c[:, 15])
    return torch.clamp(f + 0.5, min=0.0)


def rotation_transpose(q):
    """transforms.rotationMatrixTranspose, quaternion (r,x,y,z): rows [E,3,3]."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    xx, yy, zz, xy, xz, yz, rx, ry, rz = x * x, y * y, z * z, x * y, x * z, y * z, r * x, r * y, r * z
    return torch.stack([torch.stack([1 - 2 * (yy + zz), 2 * (xy + rz), 2 * (xz - ry)], -1),
                        torch.stack([2 * (xy - rz), 1 - 2 * (xx + zz), 2 * (yz + rx)], -1),
                        torch.stack([2 * (xz + ry), 2 * (yz - rx), 1 - 2 * (xx + yy)], -1)], 1)


def hit(ray_o, ray_d, pos, quat, scale, density):
    """gaussianParticle.hit for one ray against E particles (kernel degree 2): accept [E] bool, alpha [E], hit distance [E]."""
    rot_t = rotation_transpose(quat)
    giscl = 1.0 / scale
    gposc = ray_o[None, :] - pos
    gro = giscl * torch.einsum("eij,ej->ei", rot_t, gposc)
    grdu = giscl * torch.einsum("eij,j->ei", rot_t, ray_d)
    grd = grdu / grdu.norm(dim=1, keepdim=True)
    gcrod = torch.cross(grd, gro, dim=1)
    gray = (gcrod * gcrod).sum(1)
    resp = torch.exp(-0.5 * gray)
    alpha = torch.clamp(resp * density, max=MAX_ALPHA)
    accept = (resp > MIN_RESPONSE) & (alpha > MIN_ALPHA)
    grds = scale * grd * (grd * (-gro)).sum(1, keepdim=True)
    hit_t = (grds * grds).sum(1).clamp_min(1e-300).sqrt()
    return accept, alpha, hit_t


def render(K, d12, coeffs, cam_pos, lists, ranges, rays_o, rays_d, W, H):
    """The whole frame: per-particle radiance from the SH coefficients, then evalKBuffer per pixel.  Returns (fd [H,W,4], dist [H,W],
    per-particle radiance [N,3] with its gradient retained)."""
    pos, density, quat, scale = d12[:, 0:3], d12[:, 3], d12[:, 4:8], d12[:, 8:11]
    v = pos - cam_pos[None, :]
    rgb = sh_decode(coeffs, v / v.norm(dim=1, keepdim=True))
    rgb.retain_grad()
    gx = (W + 15) // 16
    fd_rows, dist_rows = [], []
    for y in range(H):
        fd_row, dist_row = [], []
        for x in range(W):
            tile = (y // 16) * gx + (x // 16)
            idx = torch.as_tensor(lists[ranges[tile, 0]:ranges[tile, 1]].astype(np.int64))
            T = torch.ones(())
            C = torch.zeros(3)
            D = torch.zeros(())
            if idx.numel():
                accept, alpha, hit_t = hit(rays_o[y, x], rays_d[y, x], pos[idx], quat[idx], scale[idx], density[idx])
                order = [int(e) for e in torch.nonzero(accept & (hit_t > 0)).flatten()]   # ray interval (0, 1e6-box exit)
                hv = hit_t.detach().numpy()
                alive = True

                def integrate(e, T, C, D):
                    w = alpha[e] * T
                    D = D + hit_t[e] * w
                    T = T * (1 - alpha[e])
                    if float(w) > 0:
                        C = C + rgb[idx[e]] * w
                    return T, C, D

                if K == 0:
                    for e in order:
                        T, C, D = integrate(e, T, C, D)
                        if float(T) < MIN_T:
                            break
                else:
                    buf = []   # pending hits, ascending in hit distance (HitParticleKBufferT::insert keeps them sorted)
                    for e in order:
                        if not alive:
                            break
                        if len(buf) == K:
                            T, C, D = integrate(buf.pop(0), T, C, D)
                            if float(T) < MIN_T:
                                alive = False
                        # insertion walks from the far end and swaps while strictly farther (:76-91): a new hit goes BEFORE pending
                        # hits of equal distance
                        pos_in = 0
                        while pos_in < len(buf) and hv[buf[pos_in]] < hv[e]:
                            pos_in += 1
                        buf.insert(pos_in, e)
                    for e in buf:
                        if not alive:
                            break
                        T, C, D = integrate(e, T, C, D)
                        if float(T) < MIN_T:
                            alive = False
            fd_row.append(torch.cat([C, (1 - T).reshape(1)]))
            dist_row.append(D)
        fd_rows.append(torch.stack(fd_row))
        dist_rows.append(torch.stack(dist_row))
    return torch.stack(fd_rows), torch.stack(dist_rows), rgb


# ---- neural harmonic features (model.feature_type = nht): the K = 0 backward with per-ray features is Slang autodiff output too -----------
#   * canonical intersection             .../slang/models/gaussianParticles.slang:181-190 (gro + grd (grd . -gro))
#   * barycentric weights, blend, sincos  .../slang/models/neuralHarmonicFeaturesParticle.slang:47-66, :117-127, :146-196
#   * integration                         :198-211 (+= features * weight), processHitParticle gutKBufferRenderer.cuh:199-225
SQ6, SQ2 = 24.0 ** 0.5 / 2, 2.0 ** 0.5
TET = torch.tensor([[SQ6, -SQ2, -1.0], [-SQ6, -SQ2, -1.0], [0.0, 24.0 ** 0.5 * 3 ** 0.5 / 2 - SQ2, -1.0], [0.0, 0.0, 3.0]])


def nht_features(feat_rows, P, ipd=12, nf=1):
    """feat_rows [48] of one particle, canonical position P [3] -> [ipd * nf * 2] sincos features."""
    e1, e2, e3 = TET[1] - TET[0], TET[2] - TET[0], TET[3] - TET[0]
    c23 = torch.linalg.cross(e2, e3)
    inv_det = 1.0 / (e1 * c23).sum()
    d = P - TET[0]
    w1 = (d * c23).sum() * inv_det
    w2 = (e1 * torch.linalg.cross(d, e3)).sum() * inv_det
    w3 = (e1 * torch.linalg.cross(e2, d)).sum() * inv_det
    w0 = 1.0 - w1 - w2 - w3
    F = feat_rows.reshape(4, ipd)
    base = F[0] * w0 + F[1] * w1 + F[2] * w2 + F[3] * w3
    out = []
    for f in range(nf):
        ang = base * (f + 1)
        out.append(torch.stack([torch.sin(ang), torch.cos(ang)], -1))           # [ipd, 2]
    return torch.stack(out, 1).reshape(-1)                                       # index k*nf*2 + f*2 + {0,1}


def render_nht(d12, feats, lists, ranges, rays_o, rays_d, W, H, nr=24, K=0):
    pos, density, quat, scale = d12[:, 0:3], d12[:, 3], d12[:, 4:8], d12[:, 8:11]
    gx = (W + 15) // 16
    fd_rows, dist_rows = [], []
    for y in range(H):
        fd_row, dist_row = [], []
        for x in range(W):
            tile = (y // 16) * gx + (x // 16)
            idx = torch.as_tensor(lists[ranges[tile, 0]:ranges[tile, 1]].astype(np.int64))
            T, C, D = torch.ones(()), torch.zeros(nr), torch.zeros(())
            if idx.numel():
                rot_t = rotation_transpose(quat[idx])
                giscl = 1.0 / scale[idx]
                gro = giscl * torch.einsum("eij,ej->ei", rot_t, rays_o[y, x][None, :] - pos[idx])
                grdu = giscl * torch.einsum("eij,j->ei", rot_t, rays_d[y, x])
                grd = grdu / grdu.norm(dim=1, keepdim=True)
                gcrod = torch.cross(grd, gro, dim=1)
                resp = torch.exp(-0.5 * (gcrod * gcrod).sum(1))
                alpha = torch.clamp(resp * density[idx], max=MAX_ALPHA)
                accept = (resp > MIN_RESPONSE) & (alpha > MIN_ALPHA)
                cg = grd * (grd * (-gro)).sum(1, keepdim=True)
                Pc = gro + cg
                grds = scale[idx] * cg
                hit_t = (grds * grds).sum(1).clamp_min(1e-300).sqrt()
                order = [int(e) for e in torch.nonzero(accept & (hit_t > 0)).flatten()]
                if K > 0:   # the sorted hit buffer (evalKBuffer, as in render above): the order in which the hits are composited
                    hv = hit_t.detach().numpy()
                    buf, seq = [], []
                    for e in order:
                        if len(buf) == K:
                            seq.append(buf.pop(0))
                        pos_in = 0
                        while pos_in < len(buf) and hv[buf[pos_in]] < hv[e]:
                            pos_in += 1
                        buf.insert(pos_in, e)
                    # (a ray that dies on a popped hit stops examining entries: the hits behind it are never reached - the break below covers
                    # both the pops and the final drain, because composited hits only ever come off the front of `seq + buf`)
                    order = seq + buf
                for e in order:
                    w = alpha[e] * T
                    D = D + hit_t[e] * w
                    T = T * (1 - alpha[e])
                    if float(w) > 0:
                        C = C + nht_features(feats[idx[e]], Pc[e]) * w
                    if float(T) < MIN_T:
                        break
            fd_row.append(torch.cat([C, (1 - T).reshape(1)]))
            dist_row.append(D)
        fd_rows.append(torch.stack(fd_row))
        dist_rows.append(torch.stack(dist_row))
    return torch.stack(fd_rows), torch.stack(dist_rows)


def case_nht(with_depth_grad, n=260, w=32, h=24, seed=5, K=0):
    scene = make_scene(n=n, width=w, height=h, median_scale=0.16, seed=seed)
    cfg = oracle.default_gut_config(k_buffer_size=K)
    feats_np = np.random.default_rng(91).uniform(-np.pi / 2, np.pi / 2, size=(n, 48))
    fwd = oracle.gut_forward_nht(cfg, scene["cam"], scene["pose_start"], scene["pose_end"], scene["density12"], feats_np, *scene["rays"], dtype=np.float64)
    T = scene["batch"]["T_to_world"][0].astype(np.float64)
    ro, rd = scene["rays"]
    rays_o = torch.as_tensor(ro[0].astype(np.float64) @ T[:3, :3].T + T[:3, 3])
    rays_d = torch.as_tensor(rd[0].astype(np.float64) @ T[:3, :3].T)
    d12 = torch.as_tensor(scene["density12"].astype(np.float64)).requires_grad_(True)
    feats = torch.as_tensor(feats_np).requires_grad_(True)
    fd, dist = render_nht(d12, feats, fwd["sorted_idx"], fwd["tile_ranges"].astype(np.int64), rays_o, rays_d, w, h, K=K)
    e_img = np.abs(fd.detach().numpy() - fwd["feat_density"]).max()
    e_dist = np.abs(dist.detach().numpy() - fwd["hit_distance"][..., 0]).max()
    assert e_img < 1e-6 and e_dist < 1e-6, (e_img, e_dist)
    rng = np.random.default_rng(23)
    g_fd = rng.normal(size=(h, w, 25))
    g_dist = rng.normal(size=(h, w)) * (0.1 if with_depth_grad else 0.0)
    ((fd * torch.as_tensor(g_fd)).sum() + (dist * torch.as_tensor(g_dist)).sum()).backward()
    print(f"nht K={K} depth_grad={with_depth_grad}: forward agrees to {e_img:.1e} / {e_dist:.1e}; hits per pixel mean {fwd['hit_count'].mean():.1f}")
    return dict(density12=scene["density12"], features=feats_np.astype(np.float32), g_fd=g_fd.astype(np.float32), g_dist=g_dist.astype(np.float32)[..., None],
                grad_density12=d12.grad.numpy().copy(), grad_features=feats.grad.numpy().copy(), n=n, w=w, h=h, seed=seed, K=K)


def case(K, with_depth_grad, n=260, w=32, h=24, seed=5):
    scene = make_scene(n=n, width=w, height=h, median_scale=0.16, seed=seed)
    cfg = oracle.default_gut_config(k_buffer_size=K)
    fwd = oracle.gut_forward(cfg, scene["cam"], scene["pose_start"], scene["pose_end"], 3, scene["density12"], scene["sph"], *scene["rays"],
                             dtype=np.float64)
    # world-space rays and the sensor position: the camera-to-world matrix applied to the camera-space rays (gutRenderer.cu:266-267)
    T = scene["batch"]["T_to_world"][0].astype(np.float64)
    ro, rd = scene["rays"]
    rays_o = torch.as_tensor(ro[0].astype(np.float64) @ T[:3, :3].T + T[:3, 3])
    rays_d = torch.as_tensor(rd[0].astype(np.float64) @ T[:3, :3].T)
    cam_pos = torch.as_tensor(T[:3, 3].copy())
    d12 = torch.as_tensor(scene["density12"].astype(np.float64)).requires_grad_(True)
    coeffs = torch.as_tensor(scene["sph"].astype(np.float64)).reshape(n, 16, 3).requires_grad_(True)
    fd, dist, rgb = render(K, d12, coeffs, cam_pos, fwd["bins"]["sorted_idx"], fwd["bins"]["tile_ranges"].astype(np.int64), rays_o, rays_d, w, h)
    # the restated forward IS the oracle's forward (and, through gut_render.npz, the reference kernels')
    e_img = np.abs(fd.detach().numpy() - fwd["feat_density"]).max()
    e_dist = np.abs(dist.detach().numpy() - fwd["hit_distance"][..., 0]).max()
    assert e_img < 1e-6 and e_dist < 1e-6, (K, e_img, e_dist)   # (pose through a float32 quaternion on the oracle's side: ~1e-8)
    rng = np.random.default_rng(17 + K)
    g_fd = rng.normal(size=(h, w, 4))
    g_dist = rng.normal(size=(h, w)) * (0.1 if with_depth_grad else 0.0)
    loss = (fd * torch.as_tensor(g_fd)).sum() + (dist * torch.as_tensor(g_dist)).sum()
    loss.backward()
    hits = fwd["hit_count"][..., 0]
    print(f"K={K} depth_grad={with_depth_grad}: forward agrees to {e_img:.1e} / {e_dist:.1e}; hits per pixel mean {hits.mean():.1f} max {hits.max():.0f}")
    return dict(density12=scene["density12"], sph=scene["sph"], g_fd=g_fd.astype(np.float32), g_dist=g_dist.astype(np.float32)[..., None],
                grad_density12=d12.grad.numpy().copy(), grad_sph=coeffs.grad.reshape(n, 48).numpy().copy(), grad_radiance=rgb.grad.numpy().copy(),
                n=n, w=w, h=h, seed=seed, K=K)


if __name__ == "__main__":
    if "--nht" in sys.argv:
        out = {}
        # (round 6: the sorted hit buffer in front of the feature integration - K = 4 and K = 16 with a hit-distance gradient)
        for name, dg, K in (("nht", False, 0), ("nht_depth", True, 0), ("nht_k4", False, 4), ("nht_k16_depth", True, 16)):
            for k, v in case_nht(dg, K=K).items():
                out[f"{name}_{k}"] = v
        np.savez_compressed(os.path.join(HERE, "autograd_gut_nht.npz"), **out)
        print("wrote tests/golden/autograd_gut_nht.npz")
        sys.exit(0)
    out = {}
    for name, K, dg in (("k0", 0, False), ("k4", 4, False), ("k16", 16, False), ("k16_depth", 16, True)):
        for k, v in case(K, dg).items():
            out[f"{name}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "autograd_gut.npz"), **out)
    print("wrote tests/golden/autograd_gut.npz")
