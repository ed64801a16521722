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

    def __init__(self, average: bool = True, group=None, local_gradient_hook=None, timed: bool = False, chunks: int = 1):
        self.average = average  # mean over views, like GradientExchange and the module docstring (keeps the single-view loss scale)
        self.group = group
        # > 1: the plugin finalises its gradients in this many particle ranges and the collectives of range i are issued under the
        # kernels of range i + 1 (reduce_packed_pipelined); 1: one exchange after the whole backward (reduce_packed)
        self.chunks = int(chunks)
        self.timer = _ExchangeTimer() if timed else None
        # local_gradient_hook(rows, first): this view's packed gradient rows [n,12] (columns 0..2 = dL/d position) of the particles
        # [first, first + n) BEFORE they are reduced - the place to take the densification statistics, which must come from the local
        # view (local_densify_stats).  ONE signature on every path: the unchunked exchanges call it once with (all N rows, 0), the
        # pipelined one once per particle range.  A hook that takes a single argument is still accepted (it is then called with the
        # rows only - fine for statistics that do not need the particle index).
        self.local_gradient_hook = local_gradient_hook
        if self.chunks > 1 and type(self) is not FactoredGradientExchange and not getattr(type(self), "PIPELINED", False):
            raise ValueError(f"{type(self).__name__} has no pipelined form: chunks must be 1 (the base FactoredGradientExchange pipelines)")

    def _call_hook(self, rows, first=0):
        hook = self.local_gradient_hook
        if hook is None:
            return
        if getattr(self, "_hook_arity_of", None) is not hook:   # (keyed on the hook object: assigning another hook later re-derives it)
            self._hook_arity_of = hook
            import inspect
            try:
                params = [p for p in inspect.signature(hook).parameters.values()
                          if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD, p.VAR_POSITIONAL)]
                self._hook_arity = 2 if (len(params) >= 2 or any(p.kind == p.VAR_POSITIONAL for p in params)) else 1
            except (TypeError, ValueError):
                self._hook_arity = 2
        if self._hook_arity == 2:
            hook(rows, first)
        else:
            hook(rows)

    def _world(self):
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    @torch.no_grad()
    def reduce_packed(self, g_density, g_radiance, positions, n_active_features, sph_degree):
        """(packed gradient [N,12] of this view, view factor [N+1,3]) -> (sum / mean over views of the packed gradient,
        SH-coefficient gradient [N, 3*(deg+1)^2] of all views).  `positions`: [N,3] or packed [N,12] particle rows."""
        from . import _abi
        world = self._world()
        scale = 1.0 / world if self.average else 1.0
        self._call_hook(g_density, 0)
        if world == 1:
            return g_density, _abi.sph_grad_from_views(g_radiance.unsqueeze(0), positions, n_active_features, sph_degree, 1.0)
        if self.timer:
            self.timer.begin(g_density.device)
        g_density, factors, payload = self.exchange(g_density, g_radiance)
        if self.timer:
            self.timer.end(payload)
        return g_density, _abi.sph_grad_from_views(factors, positions, n_active_features, sph_degree, scale)

    # ---- pipelined form -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def begin_chunk(self, g_density_rows, g_radiance_rows):
        """Issues the two collectives of one particle range WITHOUT waiting: all-reduce of the packed rows in place, gather of the view
        factors of the rows.  Returns the handle finish_chunk takes.  (RCCL: asynchronous on its own stream, ordered after what the
        compute stream holds at this point - the range's finalisation kernels; gloo: the calls complete before they return.)"""
        world = self._world()
        nccl = dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if (self.average and nccl) else dist.ReduceOp.SUM
        w_geo = dist.all_reduce(g_density_rows, op=op, group=self.group, async_op=True)
        rows = g_radiance_rows.contiguous()
        gathered = torch.empty((world,) + tuple(rows.shape), dtype=rows.dtype, device=rows.device)
        if nccl:
            w_fac = dist.all_gather_into_tensor(gathered, rows, group=self.group, async_op=True)
        else:
            gathered.zero_()
            gathered[dist.get_rank(self.group)].copy_(rows)
            w_fac = dist.all_reduce(gathered, group=self.group, async_op=True)
        return (g_density_rows, gathered, w_geo, w_fac, nccl)

    @torch.no_grad()
    def finish_chunk(self, handle):
        """-> the gathered view factors [world, rows, 3] of the range; its packed rows are reduced in place."""
        g_rows, gathered, w_geo, w_fac, nccl = handle
        w_geo.wait()
        w_fac.wait()
        if self.average and not nccl:
            g_rows.div_(float(self._world()))
        return gathered

    @torch.no_grad()
    def reduce_packed_pipelined(self, run_chunked_backward, positions, n_active_features, sph_degree):
        """run_chunked_backward(num_chunks, on_chunk) -> (g_density [N,12], g_radiance [N+1,3]) runs the plugin's chunked backward
        (gut_backward_factored_chunked), calling on_chunk(chunk, first, count, g_density, g_radiance) after each particle range.
        Range i's collectives are issued inside that call-back, i.e. before range i + 1's kernels are even launched, and are waited
        for only after the last range: on RCCL they overlap the remaining finalisation; the SH rebuild then proceeds range by range,
        each as soon as ITS factors have arrived.  Same values as reduce_packed, row for row (an all-reduce adds element-wise; views are
        added in rank order)."""
        from . import _abi
        world = self._world()
        scale = 1.0 / world if self.average else 1.0
        pending = []
        payload = [0]

        def on_chunk(chunk, first, count, g_density, g_radiance):
            rows_d, rows_r = g_density[first:first + count], g_radiance[first:first + count]
            self._call_hook(rows_d, first)   # (rows of the range, index of its first particle)
            if world > 1:
                if self.timer and not pending:
                    self.timer.begin(g_density.device)
                pending.append((first, count, self.begin_chunk(rows_d, rows_r)))
                payload[0] += rows_d.numel() * 4 + rows_r.numel() * 4

        g_density, g_radiance = run_chunked_backward(self.chunks, on_chunk)
        n = g_density.shape[0]
        g_sph = torch.empty((n, 3 * (sph_degree + 1) ** 2), dtype=torch.float32, device=g_density.device)
        if world == 1:
            _abi.sph_grad_from_views(g_radiance.unsqueeze(0), positions, n_active_features, sph_degree, 1.0, out=g_sph)
            return g_density, g_sph
        nccl = dist.get_backend(self.group) == "nccl"
        sensors = _all_gather_rows(g_radiance[n].contiguous(), self.group, nccl)     # [world, 3]: every view's sensor position
        for first, count, handle in pending:
            gathered = self.finish_chunk(handle)
            factors = torch.cat([gathered, sensors[:, None, :]], dim=1)            # [world, count + 1, 3]
            _abi.sph_grad_from_views(factors, positions[first:first + count], n_active_features, sph_degree, scale, out=g_sph[first:first + count])
        if self.timer:
            self.timer.end(payload[0])
        return g_density, g_sph

    @torch.no_grad()
    def reduce_dense(self, g_density, *others):
        """Plain in-place sum / mean of dense per-particle gradients - for features without a per-view factorisation (neural harmonic
        features: the [N,48] feature-row gradient is a sum over hits of hit-dependent weights, gut_tracer._NhtAutograd).  The packed
        gradient goes first (the local hook sees it before the reduction).  Returns the tensors."""
        world = self._world()
        self._call_hook(g_density, 0)
        tensors = [g_density, *others]
        if world == 1:
            return tensors
        nccl = dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if (self.average and nccl) else dist.ReduceOp.SUM
        if self.timer:
            self.timer.begin(g_density.device)
        works = [dist.all_reduce(t, op=op, group=self.group, async_op=True) for t in tensors]
        for w in works:
            w.wait()
        if self.timer:
            self.timer.end(sum(t.numel() * t.element_size() for t in tensors))
        if self.average and not nccl:
            torch._foreach_div_(tensors, float(world))
        return tensors

    @torch.no_grad()
    def exchange(self, g_density, g_radiance):
        """The collectives alone (no kernel of the library: runs on CPU tensors over gloo too): (reduced packed gradient [N,12], the
        views' factors [world, N+1, 3], bytes this rank handed to the collectives)."""
        world = self._world()
        nccl = dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if (self.average and nccl) else dist.ReduceOp.SUM
        w_geo = dist.all_reduce(g_density, op=op, group=self.group, async_op=True)
        factors = _all_gather_rows(g_radiance, self.group, nccl)
        w_geo.wait()
        if self.average and not nccl:
            g_density.div_(float(world))
        return g_density, factors, g_density.numel() * 4 + g_radiance.numel() * 4


def _all_gather_rows(t, group, nccl):
    """[...] of every rank -> [world, ...] on every rank."""
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if nccl:
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    else:
        # gloo (CPU-side test plumbing) has no all-gather on device tensors: every rank fills its own slice of a zeroed
        # buffer and the slices are summed, which is the same gather
        out.zero_()
        out[dist.get_rank(group)].copy_(t)
        dist.all_reduce(out, group=group)
    return out


class VisibleRowsExchange(FactoredGradientExchange):
    """The factored exchange restricted to the rows some view touched (DESIGN.md "Multi-GPU"): a view's gradient is zero on every
    particle its rays did not reach, so the masks of the non-zero rows are OR-reduced (one byte per particle on the links), every rank
    derives the same index list from the union, and the two collectives of the factored exchange run on the compacted rows — [M,12]
    all-reduced, [M+1,3] gathered — before the results are scattered back.  Pays when the union of the views leaves a good part of
    the scene untouched (a large scene seen from nearby cameras); on the bench cloud the union of 8 orbit views is nearly everything."""

    @torch.no_grad()
    def exchange(self, g_density, g_radiance):
        world = self._world()
        nccl = dist.get_backend(self.group) == "nccl"
        n = g_density.shape[0]
        touched = ((g_density != 0).any(dim=1) | (g_radiance[:n] != 0).any(dim=1)).to(torch.uint8)
        dist.all_reduce(touched, op=dist.ReduceOp.MAX, group=self.group)
        idx = touched.nonzero(as_tuple=True)[0]                      # identical on every rank (one host round trip for its length)
        m = int(idx.numel())
        geo = g_density.index_select(0, idx)
        fac = torch.cat([g_radiance.index_select(0, idx), g_radiance[n:n + 1]], dim=0)   # + the view's sensor position row
        op = dist.ReduceOp.AVG if (self.average and nccl) else dist.ReduceOp.SUM
        w_geo = dist.all_reduce(geo, op=op, group=self.group, async_op=True)
        gathered = _all_gather_rows(fac, self.group, nccl)           # [world, M+1, 3]
        w_geo.wait()
        if self.average and not nccl:
            geo.div_(float(world))
        g_density.zero_()
        g_density.index_copy_(0, idx, geo)
        factors = torch.zeros((world, n + 1, 3), dtype=g_radiance.dtype, device=g_radiance.device)
        factors[:, :n].index_copy_(1, idx, gathered[:, :m])
        factors[:, n] = gathered[:, m]
        self.last_rows = m
        return g_density, factors, n + geo.numel() * 4 + fac.numel() * 4


class ShardedGradientExchange(FactoredGradientExchange):
    """Reduce-scatter / all-gather form (DESIGN.md "Multi-GPU"): rank r owns the particle rows [r S, (r+1) S), S = ceil(N / world).
    The packed gradient is reduce-scattered (every rank receives the sum of ITS rows), the view factors travel all-to-all (every rank
    receives every view's factors of its rows: 12 B x (world-1)/world per particle on the links instead of 12 B x (world-1) for the
    all-gather), each rank rebuilds the SH gradient of its S rows only — `reduce_shard` stops there: the owner can step its rows and
    all-gather PARAMETERS — and `reduce_packed` (what the plugin's backward calls, which must hand autograd full tensors) all-gathers
    the two gradient shards.  Links per particle at 8 ranks: 42 B (reduce-scatter) + 10.5 B (all-to-all) + 210 B (all-gather of
    [S,60] gradients) against 168 B for the factored exchange: it only pays together with a sharded optimizer step."""

    def shard_rows(self, n):
        world = self._world()
        s = (n + world - 1) // world
        r = dist.get_rank(self.group) if world > 1 else 0
        return s, min(n, r * s), min(n, (r + 1) * s)   # (a rank past the end owns the empty range [n, n): N < world * (world - 1))

    @torch.no_grad()
    def exchange_shard(self, g_density, g_radiance):
        """-> (sum / mean over views of this rank's rows [S,12] (rows past N are zero), every view's factors of those rows
        [world, S+1, 3] with the views' sensor positions in row S, bytes handed to the collectives)."""
        world = self._world()
        nccl = dist.get_backend(self.group) == "nccl"
        rank = dist.get_rank(self.group)
        n = g_density.shape[0]
        s, lo, hi = self.shard_rows(n)
        pad = s * world - n
        geo = torch.cat([g_density, g_density.new_zeros((pad, g_density.shape[1]))]) if pad else g_density
        fac = torch.cat([g_radiance[:n], g_radiance.new_zeros((pad, 3))]) if pad else g_radiance[:n]
        fac = fac.contiguous()
        sensors = _all_gather_rows(g_radiance[n].contiguous(), self.group, nccl)     # [world, 3]
        if nccl:
            mine = torch.empty((s, geo.shape[1]), dtype=geo.dtype, device=geo.device)
            dist.reduce_scatter_tensor(mine, geo.contiguous(), op=dist.ReduceOp.AVG if self.average else dist.ReduceOp.SUM, group=self.group)
            views = torch.empty((world, s, 3), dtype=fac.dtype, device=fac.device)
            dist.all_to_all_single(views, fac, group=self.group)
        else:   # gloo: the same results from the collectives it has
            total = geo.clone()
            dist.all_reduce(total, group=self.group)
            mine = total[rank * s:(rank + 1) * s].clone()
            if self.average:
                mine.div_(float(world))
            views = _all_gather_rows(fac, self.group, False)[:, rank * s:(rank + 1) * s].contiguous()
        factors = torch.cat([views, sensors[:, None, :]], dim=1)
        return mine, factors, geo.numel() * 4 + fac.numel() * 4

    @torch.no_grad()
    def reduce_shard(self, g_density, g_radiance, positions, n_active_features, sph_degree):
        """-> ((lo, hi) rows owned by this rank, packed gradient of those rows [hi-lo,12], SH gradient of those rows [hi-lo, 3(deg+1)^2])."""
        from . import _abi
        world = self._world()
        n = g_density.shape[0]
        s, lo, hi = self.shard_rows(n)
        mine, factors, payload = self.exchange_shard(g_density, g_radiance)
        pos = positions[lo:hi]
        if hi - lo < s:
            pos = torch.cat([pos, pos.new_zeros((s - (hi - lo), pos.shape[1]))])
        g_sph = _abi.sph_grad_from_views(factors, pos, n_active_features, sph_degree, 1.0 / world if self.average else 1.0)
        self._payload = payload
        return (lo, hi), mine[:hi - lo], g_sph[:hi - lo]

    @torch.no_grad()
    def all_gather_rows(self, shard, n):
        """[<= S, k] of every rank (its owned rows) -> [N, k] on every rank (gradients here; parameters after a sharded optimizer step)."""
        world = self._world()
        nccl = dist.get_backend(self.group) == "nccl"
        s = (n + world - 1) // world
        if shard.shape[0] < s:
            shard = torch.cat([shard, shard.new_zeros((s - shard.shape[0],) + tuple(shard.shape[1:]))])
        return _all_gather_rows(shard.contiguous(), self.group, nccl).reshape((world * s,) + tuple(shard.shape[1:]))[:n]

    @torch.no_grad()
    def reduce_packed(self, g_density, g_radiance, positions, n_active_features, sph_degree):
        from . import _abi
        world = self._world()
        self._call_hook(g_density, 0)
        if world == 1:
            return g_density, _abi.sph_grad_from_views(g_radiance.unsqueeze(0), positions, n_active_features, sph_degree, 1.0)
        n = g_density.shape[0]
        if self.timer:
            self.timer.begin(g_density.device)
        _, geo, g_sph = self.reduce_shard(g_density, g_radiance, positions, n_active_features, sph_degree)
        both = self.all_gather_rows(torch.cat([geo, g_sph], dim=1), n)
        if self.timer:
            self.timer.end(self._payload + both.shape[1] * 4 * ((n + world - 1) // world))
        return both[:, :12].contiguous(), both[:, 12:].contiguous()


class HalfFactorsExchange(FactoredGradientExchange):
    """The factored exchange with the view factors on the links as IEEE half (6 B instead of 12 B per particle and view; at 8 ranks
    2*(7/8)*48 + (7/8)*8*6 = 126 B per particle instead of 168).  A view's radiance gradients are far below half's range (~1e-7 on a
    2 M-pixel frame), so each view scales its factors by the power of two that brings its largest magnitude to 2^14 and the exponent
    rides in the sensor row; every rank - the sender included - rebuilds the SH gradient from the ROUNDED factors, so replicas stay
    bitwise identical.  Costs 11 bits of the radiance gradient (<= 4.9e-4 of a view's largest factor): opt-in
    (GRUT_BENCH_EXCHANGE=half), not the default."""

    @torch.no_grad()
    def exchange(self, g_density, g_radiance):
        world = self._world()
        nccl = dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if (self.average and nccl) else dist.ReduceOp.SUM
        w_geo = dist.all_reduce(g_density, op=op, group=self.group, async_op=True)
        n = g_radiance.shape[0] - 1
        peak = g_radiance[:n].abs().max().clamp_min(1e-38)
        # largest magnitude in [2^e, 2^(e+1)); the exponent is held above -100 so that the scale 2^(14 - e) stays finite in fp32 - a view that
        # sees nothing (all-zero factors) or whose factors lie below 2^-100 would otherwise scale by inf and send 0 * inf = NaN to every rank
        expo = torch.floor(torch.log2(peak)).clamp(min=-100.0)
        scale = torch.exp2(14.0 - expo)                         # -> [2^14, 2^15): inside half's range, no overflow
        halves = (g_radiance[:n] * scale).to(torch.float16)
        head = torch.cat([g_radiance[n], scale.reshape(1)])     # sensor position + the scale, fp32
        heads = _all_gather_rows(head.contiguous(), self.group, nccl)                        # [world, 4]
        if nccl:
            gathered = _all_gather_rows(halves, self.group, True)                            # [world, n, 3] half
        else:   # gloo reduces neither half nor int16: the bit patterns travel widened to int32 (one non-zero contribution per element: the sum is a copy)
            gathered = _all_gather_rows(halves.view(torch.int16).to(torch.int32), self.group, False).to(torch.int16).view(torch.float16)
        w_geo.wait()
        if self.average and not nccl:
            g_density.div_(float(world))
        factors = torch.empty((world, n + 1, 3), dtype=torch.float32, device=g_radiance.device)
        factors[:, :n] = gathered.float() / heads[:, 3].reshape(world, 1, 1)
        factors[:, n] = heads[:, :3]
        return g_density, factors, g_density.numel() * 4 + halves.numel() * 2 + 16


def _p2p_exchange(sends, recvs, group):
    """One batch of point-to-point transfers: sends / recvs = [(tensor, peer)].  Over RCCL the batch is one group call whose transfers
    run concurrently on the direct xGMI links between the pairs; over gloo (CPU tests) the same calls complete one after the other."""
    ops = [dist.P2POp(dist.isend, t, peer, group=group) for t, peer in sends] + [dist.P2POp(dist.irecv, t, peer, group=group) for t, peer in recvs]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


class AllLinksExchange(FactoredGradientExchange):
    """The factored exchange as DIRECT transfers between every pair of ranks instead of ring collectives (GRUT_BENCH_EXCHANGE=alllinks).
    An MI355X node is a full mesh: 7 xGMI links per GPU, one to each peer.  A ring all-reduce / all-gather keeps one link per direction
    busy and forwards everything world - 1 times; here
      * packed gradient [N,12]: rank r owns the rows [r S, (r+1) S).  Every rank sends slice j of its local gradient straight to rank
        j (7 transfers on 7 links at once), the owner adds the views IN RANK ORDER (its own included: every owner, hence every replica,
        produces the same bits), scales, and sends its reduced slice straight to every peer;
      * view factors [N+1,3]: every rank sends its factors straight to every peer.
    Bytes per rank are those of the ring forms - 2*(7/8)*48 + 7*12 B per particle sent - but spread over 7 links, so the exposed time
    is the ring's divided by up to 7 (bench.py: exchange.predicted.all_links_ms next to .one_ring_ms).  Values: the same sums as the
    ring forms up to the order of the additions; replicas bitwise identical."""

    @torch.no_grad()
    def exchange(self, g_density, g_radiance):
        world = self._world()
        rank = dist.get_rank(self.group)
        n, cols = g_density.shape
        s = (n + world - 1) // world
        lo = [min(n, r * s) for r in range(world)]
        hi = [min(n, (r + 1) * s) for r in range(world)]
        peers = [r for r in range(world) if r != rank]
        mine = hi[rank] - lo[rank]
        # (1) slices to their owners + factors to everyone, one batch
        inbox = {r: torch.empty((mine, cols), dtype=g_density.dtype, device=g_density.device) for r in peers}
        factors = torch.empty((world,) + tuple(g_radiance.shape), dtype=g_radiance.dtype, device=g_radiance.device)
        factors[rank].copy_(g_radiance)
        fac_src = g_radiance.contiguous()
        sends = [(g_density[lo[r]:hi[r]].contiguous(), r) for r in peers if hi[r] > lo[r]] + [(fac_src, r) for r in peers]
        recvs = [(inbox[r], r) for r in peers if mine > 0] + [(factors[r], r) for r in peers]
        _p2p_exchange(sends, recvs, self.group)
        # (2) the owner's sum, views in rank order
        if mine > 0:
            total = None
            for r in range(world):
                part = g_density[lo[rank]:hi[rank]] if r == rank else inbox[r]
                total = part.clone() if total is None else total.add_(part)
            if self.average:
                total.div_(float(world))
            g_density[lo[rank]:hi[rank]].copy_(total)
        # (3) reduced slices to everyone
        outs = {r: torch.empty((hi[r] - lo[r], cols), dtype=g_density.dtype, device=g_density.device) for r in peers if hi[r] > lo[r]}
        own = g_density[lo[rank]:hi[rank]].contiguous()
        _p2p_exchange([(own, r) for r in peers if mine > 0], [(outs[r], r) for r in outs], self.group)
        for r, t in outs.items():
            g_density[lo[r]:hi[r]].copy_(t)
        sent = sum((hi[r] - lo[r]) * cols * 4 for r in peers) + len(peers) * (g_radiance.numel() * 4 + mine * cols * 4)
        return g_density, factors, sent


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
