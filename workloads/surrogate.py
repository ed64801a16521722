"""Training surrogate for the metric's second half (PSNR parity): the NeRF-Synthetic / MipNeRF360 data cannot be fetched offline, so a
synthetic teacher cloud is rendered from a ring of cameras and a perturbed copy of it is trained back through the plugin exactly the
way threedgrut/trainer.py drives it.  Used by tests/test_optim_gpu.py (oracle-rendered teacher, oracle-certified result) and by
bench.py's `psnr_surrogate` entry (HIP-rendered on both sides; the test shows the two PSNRs agree to 0.01 dB)."""
import importlib
from types import SimpleNamespace

import numpy as np

from . import synthetic as syn


def _psnr(a, b):
    return float(-10.0 * np.log10(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2) + 1e-20))


def torch_batch(batch, device):
    import torch
    return SimpleNamespace(rays_ori=torch.as_tensor(batch["rays_ori"], device=device), rays_dir=torch.as_tensor(batch["rays_dir"], device=device),
                           T_to_world=torch.as_tensor(batch["T_to_world"], device=device), T_to_world_end=None, rays_in_world_space=False,
                           intrinsics=batch.get("intrinsics"), intrinsics_OpenCVPinholeCameraModelParameters=None,
                           intrinsics_OpenCVFisheyeCameraModelParameters=None, intrinsics_FThetaCameraModelParameters=None)


def train_surrogate(method, n=100_000, w=400, h=400, views=8, steps=500, median_scale=0.01, seed=42, teacher_images=None, log=None):
    """PSNR surrogate at BASELINE config 1's scale (the NeRF-Synthetic data are not available offline): a teacher cloud of n
    Gaussians, `views` cameras on the orbit, a perturbed copy trained for `steps` iterations the way trainer.py drives the plugin
    (raw parameters -> activations -> render(train=True) -> L2 -> backward -> SelectiveAdam.step(mog_visibility)).
    Returns the trained / initial / teacher parameter sets and the batches; PSNR is the caller's business (HIP- or oracle-rendered)."""
    import torch
    d12, sph = syn.cloud_trained_like(n, seed=seed, median_scale=median_scale)
    K = syn.pinhole_intrinsics(w, h)
    ro, rd = syn.pinhole_rays(w, h, K)
    batches_np = [dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(v, n_views=views)[None], intrinsics=K) for v in range(views)]
    rng = np.random.default_rng(9)
    d12_0, sph_0 = d12.copy(), sph.copy()
    d12_0[:, 0:3] += rng.normal(size=(n, 3)).astype(np.float32) * (0.2 * median_scale)
    d12_0[:, 8:11] *= np.exp(rng.normal(size=(n, 3)) * 0.3).astype(np.float32)
    d12_0[:, 3] = np.clip(d12_0[:, 3] * np.exp(rng.normal(size=n) * 0.3).astype(np.float32), 0.01, 0.98)
    sph_0[:, :3] += rng.normal(size=(n, 3)).astype(np.float32) * 0.4
    sph_0[:, 3:] = 0
    mod = importlib.import_module("3dgrut_amd.gut_tracer" if method == "3dgut" else "3dgrut_amd.grt_tracer")
    tracer = mod.Tracer({"render": {"splat": {}}} if method == "3dgut" else {"render": {}})
    batches = [torch_batch(b, "cuda") for b in batches_np]

    def hip_images(d12_, sph_):
        g_ = syn.SimpleGaussians(d12_, sph_, requires_grad=False)
        tracer.build_acc(g_, rebuild=True)
        with torch.no_grad():
            return torch.stack([tracer.render(g_, b, train=False)["pred_features"][0] for b in batches])

    target = hip_images(d12, sph) if teacher_images is None else torch.as_tensor(teacher_images, device="cuda")
    g = syn.ActivatedGaussians(d12_0, sph_0)
    opt_mod = importlib.import_module("3dgrut_amd.optimizers")
    lrs = [1.6e-4, 5e-2, 1e-3, 5e-3, 2.5e-3, 2.5e-3 / 20]   # configs/base_gs.yaml (optimizer.params.*.lr)
    opt = opt_mod.SelectiveAdam([{"params": [p], "lr": lr} for p, lr in zip(g.parameters(), lrs)], eps=1e-15)
    for it in range(steps):
        v = it % views
        for p in g.parameters():
            p.grad = None
        tracer.build_acc(g, rebuild=True)
        out = tracer.render(g, batches[v], train=True)
        loss = ((out["pred_features"][0] - target[v]) ** 2).mean()
        loss.backward()
        opt.step(out["mog_visibility"])
    torch.cuda.synchronize()
    d12_1, sph_1 = g.packed()
    res = dict(teacher=(d12, sph), initial=(d12_0, sph_0), trained=(d12_1, sph_1), batches=batches_np, hip_images=hip_images, target=target)
    res["psnr_hip_before"] = _psnr(hip_images(d12_0, sph_0).cpu().numpy(), target.cpu().numpy())
    res["psnr_hip_after"] = _psnr(hip_images(d12_1, sph_1).cpu().numpy(), target.cpu().numpy())
    if log:
        log(f"{method}: {n} Gaussians, {views} views at {w}x{h}, {steps} steps: HIP-rendered PSNR {res['psnr_hip_before']:.2f} -> {res['psnr_hip_after']:.2f} dB")
    return res


