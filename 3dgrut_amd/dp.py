"""View-sharded data parallelism for the renderer plugins: one process per GPU, every rank renders its
own view against a full replica of the Gaussians; the single exchange step is a sum-all-reduce of the Gaussian
gradients (in place, RCCL over xGMI on the GPU box, gloo in the CPU tests).  The reference is single-GPU
(no torch.distributed anywhere), so this is new surface, kept outside the plugin proper: it wraps the gradient
tensors the unchanged trainer hooks already see.

Order of operations per iteration (SURVEY §8e caveat): (1) backward on the local view; (2) the densification
statistics of strategy/gs.py:129-139 are taken from the LOCAL, pre-reduction position gradients
(`local_densify_stats`), and the accumulators themselves are sum-reduced; (3) parameter gradients are reduced
(mean over views keeps the single-view loss scale); (4) `mog_visibility` is OR-reduced for SelectiveAdam.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradientExchange:
    """Sum- (or mean-) all-reduce of the `.grad` of a fixed list of parameters, IN PLACE: the five Gaussian gradient
    tensors are reduced where they lie, as collectives issued back to back on the communication stream (RCCL runs them
    concurrently with whatever is still queued on the compute stream; nothing is packed into or out of a staging buffer —
    at 1M Gaussians that staging alone would move 2 x 236 MB per step)."""

    def __init__(self, params, average: bool = True, group=None, timed: bool = False):
        self.params = list(params)
        self.average = average
        self.group = group
        self.timer = _ExchangeTimer() if timed else None

    @torch.no_grad()
    def reduce(self):
        """Missing grads count as zero (every rank must enter the same collectives)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        world = dist.get_world_size(self.group)
        grads = []
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            elif not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
            grads.append(p.grad)
        # RCCL averages inside the collective; gloo (CPU tests) has no AVG
        native_avg = self.average and dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if native_avg else dist.ReduceOp.SUM
        if self.timer:
            self.timer.begin(grads[0].device)
        works = [dist.all_reduce(g, op=op, group=self.group, async_op=True) for g in grads]
        for w in works:
            w.wait()
        if self.timer:
            self.timer.end(sum(g.numel() * g.element_size() for g in grads))
        if self.average and not native_avg:
            torch._foreach_div_(grads, float(world))


class _ExchangeTimer:
    """Device time of the exchange step, per rank (bench.py reports the slowest rank's average): events on the stream the collectives
    are issued from / waited on, read back lazily."""

    def __init__(self):
        self.pairs, self.bytes = [], 0

    def begin(self, device):
        self._start = torch.cuda.Event(enable_timing=True) if device.type == "cuda" else None
        if self._start is not None:
            self._start.record()

    def end(self, payload_bytes):
        if self._start is not None:
            stop = torch.cuda.Event(enable_timing=True)
            stop.record()
            self.pairs.append((self._start, stop))
        self.bytes = payload_bytes

    def collect(self):
        """(average ms per exchange, payload bytes of one exchange on this rank) since the last call."""
        if not self.pairs:
            return None, self.bytes
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.pairs) / len(self.pairs)
        self.pairs = []
        return ms, self.bytes


class FactoredGradientExchange:
    """The 3DGUT plugin's exchange step, sized for xGMI (set `tracer.gradient_exchange = FactoredGradientExchange()`; the
    plugin's backward then calls `reduce_packed` before it unpacks anything).

    Per step and rank the all-reduce of the five Gaussian gradient tensors moves [N,59] floats — 236 B per particle, 192 B of
    which are the SH-coefficient gradient.  Per view that gradient is an outer product: (SH basis at the particle's view
    direction) x (3-float radiance gradient behind the clamp).  So instead:
      * the packed geometric gradient [N,12] (position, density, rotation, scale) is all-reduced where it lies — one
        collective instead of four;
      * every rank contributes its view factor [N+1,3] (`gut_backward_factored`: radiance gradients + the view's sensor
        position) to one all-gather — 12 B per particle and view on the links;
      * every rank rebuilds sum_v basis(dir_v) x g_v locally (`grut_sph_grad_from_views`, one pass that writes [N,48] once).
    At 8 ranks a ring moves 2*(7/8)*236 = 413 B per particle for the plain all-reduce, 2*(7/8)*48 + (7/8)*8*12 = 168 B this
    way.  Views are added in rank order on every rank, so replicas stay bitwise identical, like after an all-reduce."""

    def __init__(self, average: bool = True, group=None, local_gradient_hook=None, timed: bool = False):
        self.average = average  # mean over views, like GradientExchange and the module docstring (keeps the single-view loss scale)
        self.group = group
        self.timer = _ExchangeTimer() if timed else None
        # called with this view's packed gradient [N,12] (columns 0..2 = dL/d position) BEFORE it is reduced: the place to take
        # the densification statistics, which must come from the local view (local_densify_stats)
        self.local_gradient_hook = local_gradient_hook

    def _world(self):
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    @torch.no_grad()
    def reduce_packed(self, g_density, g_radiance, positions, n_active_features, sph_degree):
        """(packed gradient [N,12] of this view, view factor [N+1,3]) -> (sum / mean over views of the packed gradient,
        SH-coefficient gradient [N, 3*(deg+1)^2] of all views).  `positions`: [N,3] or packed [N,12] particle rows."""
        from . import _abi
        world = self._world()
        scale = 1.0 / world if self.average else 1.0
        if self.local_gradient_hook is not None:
            self.local_gradient_hook(g_density)
        if world == 1:
            return g_density, _abi.sph_grad_from_views(g_radiance.unsqueeze(0), positions, n_active_features, sph_degree, 1.0)
        nccl = dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if (self.average and nccl) else dist.ReduceOp.SUM
        if self.timer:
            self.timer.begin(g_density.device)
        w_geo = dist.all_reduce(g_density, op=op, group=self.group, async_op=True)
        factors = torch.empty((world,) + tuple(g_radiance.shape), dtype=g_radiance.dtype, device=g_radiance.device)
        if nccl:
            w_rad = dist.all_gather_into_tensor(factors, g_radiance.contiguous(), group=self.group, async_op=True)
        else:
            # gloo (CPU-side test plumbing) has no all-gather on device tensors: every rank fills its own slice of a zeroed
            # buffer and the slices are summed, which is the same gather
            factors.zero_()
            factors[dist.get_rank(self.group)].copy_(g_radiance)
            w_rad = dist.all_reduce(factors, group=self.group, async_op=True)
        w_geo.wait()
        w_rad.wait()
        if self.timer:
            self.timer.end(g_density.numel() * 4 + g_radiance.numel() * 4)
        if self.average and not nccl:
            g_density.div_(float(world))
        return g_density, _abi.sph_grad_from_views(factors, positions, n_active_features, sph_degree, scale)


@torch.no_grad()
def local_densify_stats(positions_grad: torch.Tensor, positions: torch.Tensor, camera_position: torch.Tensor):
    """Per-view densification statistic of strategy/gs.py:129-139: ||dL/dmu|| * dist(mu, camera) / 2 where the view's
    gradient is non-zero, plus the visit mask.  Must be computed BEFORE the gradients are reduced."""
    dist_to_cam = (positions - camera_position.view(1, 3)).norm(dim=1, keepdim=True)
    norm = positions_grad.norm(dim=1, keepdim=True) * dist_to_cam * 0.5
    mask = (positions_grad != 0).any(dim=1, keepdim=True)
    return torch.where(mask, norm, torch.zeros_like(norm)), mask.to(norm.dtype)


@torch.no_grad()
def reduce_densify_accumulators(grad_norm_accum: torch.Tensor, denom: torch.Tensor, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        both = torch.cat([grad_norm_accum.reshape(-1), denom.reshape(-1)])
        dist.all_reduce(both, group=group)
        n = grad_norm_accum.numel()
        grad_norm_accum.copy_(both[:n].view_as(grad_norm_accum))
        denom.copy_(both[n:].view_as(denom))


@torch.no_grad()
def reduce_visibility(mog_visibility: torch.Tensor, group=None) -> torch.Tensor:
    """Logical OR over ranks of the plugin's `mog_visibility` (a float tensor holding int bit patterns: `.bool()`)."""
    v = mog_visibility.bool().to(torch.int32)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
    return v.bool()


def shard_views(num_views: int, rank: int, world: int):
    """View indices rendered by `rank` in one iteration of `num_views` views (round-robin, no collective)."""
    return list(range(rank, num_views, world))
