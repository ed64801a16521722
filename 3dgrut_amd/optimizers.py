"""SelectiveAdam — the visibility-masked Adam of the reference trainer (threedgrut/optimizers/__init__.py:38-124),
backed by one HIP launch for all parameter groups (csrc/optim.hip) instead of one CUDA launch per group.

Same surface: `SelectiveAdam(params, lr, betas, eps)`; `step(visibility)` where `visibility` is the tracers'
`mog_visibility` ([N,1] float tensor holding an int bit pattern, or a bool / int tensor); one tensor per param group;
no bias correction; rows that were not visible keep parameter and moments.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi


class SelectiveAdam(torch.optim.Adam):
    def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-08):
        super().__init__(params=params, lr=lr, eps=eps, betas=betas)
        self._lib = _abi.load_library()

    @torch.no_grad()
    def step(self, visibility):
        groups, keep = [], []
        n_rows, device = None, None
        for group in self.param_groups:
            assert len(group["params"]) == 1, "More than one tensor in group is not supported"
            param = group["params"][0]
            if param.grad is None:
                continue
            if not param.is_cuda:
                raise RuntimeError("3dgrut_amd.SelectiveAdam updates GPU tensors only (there is no CPU fallback)")
            if param.dtype != torch.float32 or not param.is_contiguous():
                raise RuntimeError("3dgrut_amd.SelectiveAdam needs contiguous float32 parameters")
            state = self.state[param]
            if len(state) == 0:  # lazy state initialisation, as optimizers/__init__.py:101-105
                state["step"] = torch.tensor(0.0, dtype=torch.float32)
                state["exp_avg"] = torch.zeros_like(param, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(param, memory_format=torch.preserve_format)
            if param.numel() == 0:
                continue
            rows = int(param.shape[0])
            if n_rows is None:
                n_rows, device = rows, param.device
            elif rows != n_rows:
                raise RuntimeError(f"SelectiveAdam: parameter groups disagree on the number of rows ({rows} vs {n_rows})")
            grad = param.grad.contiguous()
            beta1, beta2 = group["betas"]
            g = _abi.GrutAdamGroup()
            g.param, g.grad = param.data_ptr(), grad.data_ptr()
            g.exp_avg, g.exp_avg_sq = state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr()
            g.row_width = param.numel() // rows
            g.lr, g.beta1, g.beta2, g.eps = float(group["lr"]), float(beta1), float(beta2), float(group["eps"])
            groups.append(g)
            keep.append(grad)
        if not groups:
            return
        vis = visibility.reshape(-1)
        if vis.numel() != n_rows:
            raise RuntimeError(f"SelectiveAdam: visibility has {vis.numel()} entries for {n_rows} rows")
        if vis.device != device:
            vis = vis.to(device)
        if vis.dtype == torch.bool:
            kind = _abi.VIS_BOOL_U8
        elif vis.dtype == torch.float32:
            kind = _abi.VIS_FLOAT_BITS  # `.bool()` of the tracers' float tensor: any non-zero bit pattern (denormals included)
        elif vis.dtype == torch.int32:
            kind = _abi.VIS_INT32
        else:
            vis, kind = vis.bool(), _abi.VIS_BOOL_U8
        vis = vis.contiguous()
        arr = (_abi.GrutAdamGroup * len(groups))(*groups)
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        _abi.check(self._lib.grut_selective_adam_update(stream, arr, len(groups), n_rows, C.c_void_p(vis.data_ptr()), kind),
                   "grut_selective_adam_update")
