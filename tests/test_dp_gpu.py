"""GPU, world_size 2: the plugin-level exchange step (3dgrut_amd/dp.FactoredGradientExchange inside the 3DGUT backward).

Two ranks share the one GPU of the test box and talk over gloo (RCCL refuses two ranks on one device); what is checked is
the data path — all-reduce of the packed gradient, gather of the view factors, local rebuild of the SH gradient — against the
sum of two single-process backward passes."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from scenes import make_scene, torch_batch

pytestmark = pytest.mark.gpu
SCENE = dict(n=2500, width=80, height=48, median_scale=0.06)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _backward(view, exchange):
    syn = importlib.import_module("3dgrut_amd.synthetic")
    gt = importlib.import_module("3dgrut_amd.gut_tracer")
    scene = make_scene(view=view, **SCENE)
    tr = gt.Tracer({"render": {"splat": {}}})
    tr.gradient_exchange = exchange
    g = syn.SimpleGaussians(scene["density12"], scene["sph"])
    out = tr.render(g, torch_batch(scene["batch"], "cuda"), train=True)
    g_fd = torch.as_tensor(syn.upstream_grads(SCENE["width"], SCENE["height"])[0] * SCENE["width"] * SCENE["height"], device="cuda")
    fd = torch.cat([out["pred_features"], out["pred_opacity"]], dim=-1)[0]
    (fd * g_fd).sum().backward()
    torch.cuda.synchronize()
    return g.grads_packed()


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = importlib.import_module("3dgrut_amd.dp")
    out[rank] = _backward(2 * rank + 1, dp.FactoredGradientExchange(average=True))
    dist.barrier()
    dist.destroy_process_group()


def test_factored_exchange_world2_on_one_gpu():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (gd0, gs0), (gd1, gs1) = out[0], out[1]
    assert np.array_equal(gd0, gd1) and np.array_equal(gs0, gs1)          # replicas identical after the exchange
    a, b = _backward(1, None), _backward(3, None)
    want_gd = 0.5 * (a[0].astype(np.float64) + b[0].astype(np.float64))
    want_gs = 0.5 * (a[1].astype(np.float64) + b[1].astype(np.float64))
    assert np.abs(gd0 - want_gd).max() <= 2e-6 * np.abs(want_gd).max()
    assert np.abs(gs0 - want_gs).max() <= 2e-6 * np.abs(want_gs).max()
    assert np.abs(want_gs).max() > 0
