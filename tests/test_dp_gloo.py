"""CPU, world_size 2 over gloo: the N>1 path's exchange step (3dgrut_amd/dp.py) is correct by construction."""
import importlib
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = importlib.import_module("3dgrut_amd.dp")
    torch.manual_seed(0)
    n = 257
    params = [torch.zeros(n, 3), torch.zeros(n, 4), torch.zeros(n, 3), torch.zeros(n, 1), torch.zeros(n, 48)]
    g = torch.Generator().manual_seed(100 + rank)
    for i, p in enumerate(params):
        if not (rank == 1 and i == 3):  # rank 1 never produced a density gradient: counts as zero
            p.grad = torch.randn(p.shape, generator=g)
    local = [None if p.grad is None else p.grad.clone() for p in params]
    cam = torch.tensor([0.0, 0.0, float(rank + 1)])
    pos = torch.randn(n, 3, generator=torch.Generator().manual_seed(5))
    stat, mask = dp.local_densify_stats(params[0].grad, pos, cam)
    acc, den = stat.clone(), mask.clone()
    dp.reduce_densify_accumulators(acc, den)
    ex = dp.GradientExchange(params, average=True)
    ex.reduce()
    vis = torch.zeros(n, 1)
    vis[rank::2] = 1.4e-45  # int bit pattern 1
    red_vis = dp.reduce_visibility(vis)
    out[rank] = dict(grads=[p.grad.numpy().copy() for p in params], local=[None if l is None else l.numpy() for l in local],
                     stat=stat.numpy(), acc=acc.numpy(), den=den.numpy(), vis=red_vis.numpy(), views=dp.shard_views(5, rank, world))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_exchange_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    for i in range(5):
        a = r0["local"][i]
        b = r1["local"][i] if r1["local"][i] is not None else np.zeros_like(a)
        np.testing.assert_allclose(r0["grads"][i], (a + b) / 2, rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(r0["grads"][i], r1["grads"][i])  # identical on every rank
    np.testing.assert_allclose(r0["acc"], r0["stat"] + r1["stat"], rtol=1e-6)
    np.testing.assert_array_equal(r0["den"], np.full_like(r0["den"], 2.0))
    assert r0["vis"].all() and r1["vis"].all()
    assert r0["views"] == [0, 2, 4] and r1["views"] == [1, 3]


def _variant_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = importlib.import_module("3dgrut_amd.dp")
    n = 257   # odd on purpose: the shards are padded
    g = torch.Generator().manual_seed(7 + rank)
    geo = torch.randn(n, 12, generator=g)
    fac = torch.randn(n + 1, 3, generator=g)          # row n: this view's sensor position
    seen = torch.rand(n, generator=g) < 0.4           # rows this view touched
    geo[~seen] = 0
    fac[:n][~seen] = 0
    res = dict(geo=geo.numpy().copy(), fac=fac.numpy().copy())
    ref_geo, ref_fac, _ = dp.FactoredGradientExchange(average=True).exchange(geo.clone(), fac.clone())
    res["ref_geo"], res["ref_fac"] = ref_geo.numpy(), ref_fac.numpy()
    vis = dp.VisibleRowsExchange(average=True)
    v_geo, v_fac, v_bytes = vis.exchange(geo.clone(), fac.clone())
    res["vis_geo"], res["vis_fac"], res["vis_rows"], res["vis_bytes"] = v_geo.numpy(), v_fac.numpy(), vis.last_rows, v_bytes
    sh = dp.ShardedGradientExchange(average=True)
    s, lo, hi = sh.shard_rows(n)
    mine, factors, _ = sh.exchange_shard(geo.clone(), fac.clone())
    res["shard"] = (s, lo, hi)
    res["sh_geo"], res["sh_fac"] = mine.numpy(), factors.numpy()
    res["sh_full"] = sh.all_gather_rows(mine[:hi - lo], n).numpy()
    # round 5: view factors as scaled halves; direct pairwise transfers instead of ring collectives
    hf_geo, hf_fac, hf_bytes = dp.HalfFactorsExchange(average=True).exchange(geo.clone(), fac.clone() * 1e-7)   # (gradients of a 2 M-pixel frame: ~1e-7)
    res["hf_geo"], res["hf_fac"], res["hf_bytes"] = hf_geo.numpy(), hf_fac.numpy(), hf_bytes
    # (round 6, advisor) a view that sees nothing - all-zero factors on rank 1 - must not turn into NaN on every rank (scale 2^(14 - e) overflowed)
    zf = fac.clone() * 1e-7
    if rank == 1:
        zf[:n] = 0
    z_geo, z_fac, _ = dp.HalfFactorsExchange(average=True).exchange(geo.clone(), zf)
    res["hfz_fac"], res["hfz_in"] = z_fac.numpy(), zf.numpy().copy()
    al_geo, al_fac, al_bytes = dp.AllLinksExchange(average=True).exchange(geo.clone(), fac.clone())
    res["al_geo"], res["al_fac"], res["al_bytes"] = al_geo.numpy(), al_fac.numpy(), al_bytes
    # fewer particles than ranks x (ranks - 1): the last rank owns the empty range [n, n) (ShardedGradientExchange.shard_rows clamps)
    tiny_geo, tiny_fac = geo[:1].clone(), torch.cat([fac[:1], fac[n:]]).clone()
    res["tiny_in"] = (tiny_geo.numpy().copy(), tiny_fac.numpy().copy())
    res["tiny_shard"] = sh.shard_rows(1)
    t_mine, t_factors, _ = sh.exchange_shard(tiny_geo.clone(), tiny_fac.clone())
    res["tiny_mine"], res["tiny_factors"] = t_mine.numpy(), t_factors.numpy()
    s1, lo1, hi1 = sh.shard_rows(1)
    res["tiny_full"] = sh.all_gather_rows(t_mine[:hi1 - lo1], 1).numpy()
    al1_geo, al1_fac, _ = dp.AllLinksExchange(average=True).exchange(tiny_geo.clone(), tiny_fac.clone())
    res["tiny_al_geo"], res["tiny_al_fac"] = al1_geo.numpy(), al1_fac.numpy()
    for cls in (dp.VisibleRowsExchange, dp.ShardedGradientExchange, dp.HalfFactorsExchange, dp.AllLinksExchange):
        try:
            cls(chunks=4)
            res["chunks_rejected"] = False
        except ValueError:
            res.setdefault("chunks_rejected", True)
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_variants_world2():
    """GRUT_BENCH_EXCHANGE=visible / sharded (dp.VisibleRowsExchange, dp.ShardedGradientExchange): their collectives deliver what the
    factored exchange delivers — the mean packed gradient and every view's factors — on the touched rows only, resp. shard by shard."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_variant_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = [out[0], out[1]]
    n = 257
    mean_geo = (r[0]["geo"] + r[1]["geo"]) / 2
    facs = np.stack([r[0]["fac"], r[1]["fac"]])
    for k in range(world):
        np.testing.assert_allclose(r[k]["ref_geo"], mean_geo, rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(r[k]["ref_fac"], facs)
        # visible rows: same results, fewer rows on the wire
        np.testing.assert_allclose(r[k]["vis_geo"], mean_geo, rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(r[k]["vis_fac"], facs)
        touched = int(((r[0]["geo"] != 0).any(1) | (r[1]["geo"] != 0).any(1) | (r[0]["fac"][:n] != 0).any(1) | (r[1]["fac"][:n] != 0).any(1)).sum())
        assert r[k]["vis_rows"] == touched and touched < 0.8 * n
        assert r[k]["vis_bytes"] == n + touched * 48 + (touched + 1) * 12
        # shards: this rank's rows of the mean, every view's factors of those rows + the sensor positions
        s, lo, hi = r[k]["shard"]
        assert (s, lo) == (129, 129 * k) and hi == min(n, 129 * (k + 1))
        np.testing.assert_allclose(r[k]["sh_geo"][:hi - lo], mean_geo[lo:hi], rtol=1e-6, atol=1e-7)
        assert not r[k]["sh_geo"][hi - lo:].any()
        np.testing.assert_array_equal(r[k]["sh_fac"][:, :hi - lo], facs[:, lo:hi])
        np.testing.assert_array_equal(r[k]["sh_fac"][:, s], facs[:, n])
        np.testing.assert_allclose(r[k]["sh_full"], mean_geo, rtol=1e-6, atol=1e-7)
        # half factors: the packed gradient as before; every view's factors within half precision of ITS largest magnitude, sensor rows exact
        np.testing.assert_allclose(r[k]["hf_geo"], mean_geo, rtol=1e-6, atol=1e-7)
        for v in range(world):
            want = facs[v, :n] * 1e-7
            assert np.abs(r[k]["hf_fac"][v, :n] - want).max() <= 4.9e-4 * np.abs(want).max()
            assert np.abs(r[k]["hf_fac"][v, :n]).max() > 0.5 * np.abs(want).max()
        np.testing.assert_allclose(r[k]["hf_fac"][:, n], facs[:, n] * 1e-7, rtol=1e-6)
        assert r[k]["hf_bytes"] == n * 48 + n * 6 + 16
        # the all-zero view of rank 1: finite everywhere, its rows exactly zero, rank 0's view as precise as before
        assert np.isfinite(r[k]["hfz_fac"]).all()
        assert not r[k]["hfz_fac"][1, :n].any()
        want0 = r[0]["hfz_in"][:n]
        assert np.abs(r[k]["hfz_fac"][0, :n] - want0).max() <= 4.9e-4 * np.abs(want0).max()
        # direct transfers: same sums (up to the order of two additions: exact here), same factors
        np.testing.assert_allclose(r[k]["al_geo"], mean_geo, rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(r[k]["al_fac"], facs)
        # one particle on two ranks
        assert r[k]["tiny_shard"] == ((1, 0, 1) if k == 0 else (1, 1, 1))
        tiny_mean = (r[0]["tiny_in"][0] + r[1]["tiny_in"][0]) / 2
        if k == 0:
            np.testing.assert_allclose(r[k]["tiny_mine"][:1], tiny_mean, rtol=1e-6, atol=1e-7)
            np.testing.assert_array_equal(r[k]["tiny_factors"][:, 0], np.stack([r[0]["tiny_in"][1][0], r[1]["tiny_in"][1][0]]))
        np.testing.assert_array_equal(r[k]["tiny_factors"][:, 1], np.stack([r[0]["tiny_in"][1][1], r[1]["tiny_in"][1][1]]))   # sensor rows
        np.testing.assert_allclose(r[k]["tiny_full"], tiny_mean, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(r[k]["tiny_al_geo"], tiny_mean, rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(r[k]["tiny_al_fac"], np.stack([r[0]["tiny_in"][1], r[1]["tiny_in"][1]]))
        assert r[k]["chunks_rejected"] is True
    # replicas are bitwise identical for every variant
    for key in ("hf_geo", "hf_fac", "al_geo", "al_fac", "tiny_full", "tiny_al_geo"):
        np.testing.assert_array_equal(r[0][key], r[1][key])


def _pipelined_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = importlib.import_module("3dgrut_amd.dp")
    n = 1000
    g = torch.Generator().manual_seed(11 + rank)
    geo = torch.randn(n, 12, generator=g)
    fac = torch.randn(n + 1, 3, generator=g)
    ref_geo, ref_fac, _ = dp.FactoredGradientExchange(average=True).exchange(geo.clone(), fac.clone())
    # the plugin's chunked backward, emulated: ranges at multiples of 128 become complete one after the other and the call-back issues
    # their collectives at once; everything is waited for only after the last range (the order reduce_packed_pipelined uses)
    ex = dp.FactoredGradientExchange(average=True, chunks=4)
    g_density, g_radiance = geo.clone(), fac.clone()
    per = ((n + ex.chunks - 1) // ex.chunks + 127) & ~127
    order, pending = [], []
    for first in range(0, n, per):
        count = min(per, n - first)
        order.append(("issue", first))
        pending.append((first, count, ex.begin_chunk(g_density[first:first + count], g_radiance[first:first + count])))
    gathered = torch.empty((world, n, 3))
    for first, count, h in pending:
        order.append(("finish", first))
        gathered[:, first:first + count] = ex.finish_chunk(h)
    out[rank] = dict(ref_geo=ref_geo.numpy(), ref_fac=ref_fac.numpy(), geo=g_density.numpy(), fac=gathered.numpy(), order=order, per=per)
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_exchange_equals_the_one_piece_exchange_world2():
    """dp.FactoredGradientExchange(chunks = 4): the collectives of particle range i are issued as soon as the range is final (under the
    kernels of range i + 1 on the GPU) and waited for at the end.  Every range's all-reduce / gather must deliver exactly the rows the
    one-piece exchange delivers - bit for bit, on both replicas - whatever the number of ranges in flight."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_pipelined_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = [out[0], out[1]]
    assert r[0]["per"] == 256 and [o for o in r[0]["order"] if o[0] == "issue"] == [("issue", f) for f in (0, 256, 512, 768)]
    assert r[0]["order"][:4] == [("issue", f) for f in (0, 256, 512, 768)], "all ranges are issued before the first is waited for"
    for k in range(world):
        np.testing.assert_array_equal(r[k]["geo"], r[k]["ref_geo"])
        np.testing.assert_array_equal(r[k]["fac"], r[k]["ref_fac"][:, :1000])
    np.testing.assert_array_equal(r[0]["geo"], r[1]["geo"])
    np.testing.assert_array_equal(r[0]["fac"], r[1]["fac"])


def test_local_gradient_hook_has_one_signature_on_every_path(monkeypatch):
    """the pipelined exchange called the hook with (rows, first), the one-piece exchanges with (rows) - a documented
    one-argument hook raised as soon as chunks > 1.  Now every path calls hook(rows, first); one-argument hooks are still accepted."""
    dp = importlib.import_module("3dgrut_amd.dp")
    abi = importlib.import_module("3dgrut_amd._abi")
    monkeypatch.setattr(abi, "sph_grad_from_views", lambda factors, positions, n_active, deg, scale, out=None:
                        (out if out is not None else torch.zeros(factors.shape[1] - 1, 3 * (deg + 1) ** 2)))
    n = 1000
    geo, fac, pos = torch.randn(n, 12), torch.randn(n + 1, 3), torch.randn(n, 3)

    def chunked_backward(num_chunks, on_chunk):
        per = ((n + num_chunks - 1) // num_chunks + 127) & ~127
        for k, first in enumerate(range(0, n, per)):
            on_chunk(k, first, min(per, n - first), geo, fac)
        return geo, fac

    for chunks in (1, 4):
        two, one = [], []
        ex2 = dp.FactoredGradientExchange(local_gradient_hook=lambda rows, first: two.append((first, rows.shape[0])), chunks=chunks)
        ex1 = dp.FactoredGradientExchange(local_gradient_hook=lambda rows: one.append(rows.shape[0]), chunks=chunks)
        for ex in (ex2, ex1):
            if chunks == 1:
                ex.reduce_packed(geo, fac, pos, 3, 3)
                ex.reduce_dense(geo)
            else:
                ex.reduce_packed_pipelined(chunked_backward, pos, 3, 3)
        if chunks == 1:
            assert two == [(0, n), (0, n)] and one == [n, n]
        else:
            assert two == [(0, 256), (256, 256), (512, 256), (768, 232)] and one == [256, 256, 256, 232]
