"""View-sharded data parallelism for the renderer plugins: one process per GPU, every rank renders its
own view against a full replica of the Gaussians; the single exchange step is a sum-all-reduce of the Gaussian
gradients (one fused buffer, RCCL over xGMI on the GPU box, gloo in the CPU tests).  The reference is single-GPU
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
    """Fused all-reduce of a fixed list of gradient tensors through one flat, persistent buffer."""

    def __init__(self, params, average: bool = True, group=None):
        self.params = list(params)
        self.average = average
        self.group = group
        self.sizes = [p.numel() for p in self.params]
        self.flat = None

    def _ensure(self, like: torch.Tensor):
        n = sum(self.sizes)
        if self.flat is None or self.flat.numel() != n or self.flat.device != like.device:
            self.flat = torch.empty(n, dtype=torch.float32, device=like.device)

    @torch.no_grad()
    def reduce(self):
        """Sum (or mean) the `.grad` of every parameter over all ranks; missing grads count as zero."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        self._ensure(self.params[0])
        off = 0
        for p, n in zip(self.params, self.sizes):
            seg = self.flat[off:off + n]
            if p.grad is None:
                seg.zero_()
            else:
                seg.copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            self.flat.div_(dist.get_world_size(self.group))
        off = 0
        for p, n in zip(self.params, self.sizes):
            seg = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = seg.clone()
            else:
                p.grad.copy_(seg)
            off += n


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
