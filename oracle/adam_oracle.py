"""CPU restatement of the reference's SelectiveAdam update (threedgrut/optimizers/optimizers.cu:49-77) in numpy fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/golden/adam.npz, which the reference's own kernel
produced (oracle/ref/ref_adam.cpp runs it thread by thread on the host; tests/golden/make_golden.py).
"""
import numpy as np

F = np.float32


def selective_adam_update(param, grad, exp_avg, exp_avg_sq, visibility, lr, b1, b2, eps):
    """-> (param, exp_avg, exp_avg_sq) after one step; inputs [N, ...] float32, visibility [N] bool.

    optimizers.cu:57-76: rows with visibility false return before touching anything; for the others, in fp32 and in
    this operation order,  m = b1*m + (1-b1)*g;  v = b2*v + ((1-b2)*g)*g;  step = -lr*m/(sqrt(v)+eps);  p += step."""
    p = np.array(param, F, copy=True)
    g = np.asarray(grad, F)
    m = np.array(exp_avg, F, copy=True)
    v = np.array(exp_avg_sq, F, copy=True)
    vis = np.asarray(visibility).reshape(-1).astype(bool)
    lr, b1, b2, eps = F(lr), F(b1), F(b2), F(eps)
    one = F(1.0)
    gm, mm, vm, pm = g[vis], m[vis], v[vis], p[vis]
    mm = (b1 * mm + (one - b1) * gm).astype(F)
    vm = (b2 * vm + ((one - b2) * gm).astype(F) * gm).astype(F)
    step = ((-lr * mm).astype(F) / (np.sqrt(vm).astype(F) + eps)).astype(F)
    p[vis] = (pm + step).astype(F)
    m[vis] = mm
    v[vis] = vm
    return p, m, v
