"""Integer stages on the GPU: scan and radix sort must be bit-exact against numpy (stable order)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("n", [1, 63, 64, 2047, 2048, 2049, 100_000, 1_234_567])
def test_inclusive_scan(grut_lib, n):
    import torch
    rng = np.random.default_rng(n)
    x = rng.integers(0, 50, n, dtype=np.uint32)
    d = torch.as_tensor(x.view(np.int32), device="cuda")
    o = torch.zeros_like(d)
    sb = int(grut_lib.grut_scan_scratch_bytes(n))
    scratch = torch.zeros(sb, dtype=torch.uint8, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert grut_lib.grut_inclusive_scan_u32(s, n, _p(d), _p(o), _p(scratch), sb) == 0
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy().view(np.uint32), np.cumsum(x, dtype=np.uint64).astype(np.uint32))


@pytest.mark.parametrize("n,bits", [(1, 32), (100, 32), (4096, 32), (4097, 12), (300_000, 13), (1_000_003, 32), (2_000_000, 8)])
def test_sort_pairs_stable(grut_lib, n, bits):
    import torch
    rng = np.random.default_rng(n + bits)
    hi = (1 << bits) - 1
    keys = rng.integers(0, hi + 1, n, dtype=np.uint64).astype(np.uint32)
    if n > 1000:  # many duplicates -> exercises stability
        keys[: n // 2] = keys[: n // 2] & np.uint32(0xFF)
    vals = np.arange(n, dtype=np.uint32)
    k = torch.as_tensor(keys.view(np.int32), device="cuda")
    v = torch.as_tensor(vals.view(np.int32), device="cuda")
    kt, vt = torch.zeros_like(k), torch.zeros_like(v)
    sb = int(grut_lib.grut_sort_scratch_bytes(n))
    scratch = torch.zeros(sb, dtype=torch.uint8, device="cuda")
    ok, ov = C.c_void_p(), C.c_void_p()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert grut_lib.grut_sort_pairs_u32(s, n, 0, bits, _p(k), _p(v), _p(kt), _p(vt), _p(scratch), sb, C.byref(ok), C.byref(ov)) == 0
    torch.cuda.synchronize()
    sk = (k if ok.value == k.data_ptr() else kt).cpu().numpy().view(np.uint32)
    sv = (v if ov.value == v.data_ptr() else vt).cpu().numpy().view(np.uint32)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(sk, keys[order])
    assert np.array_equal(sv, vals[order])
