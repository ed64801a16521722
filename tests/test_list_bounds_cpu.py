"""The hit-distance bounds behind the 3DGRT packet lists (3dgrut_amd/csrc/grt_kernels.hip: bin_particle, packet_bounds),
checked as mathematics in float64 against brute force: for random anisotropic proxies and random cones of rays,

  * every ray that touches the proxy's unit box has its hit distance t (the parameter of the point closest to the centre in the
    proxy's metric, gaussianParticles.cuh:449-466) inside [lo, hi] of packet_bounds and inside [key, ub] of bin_particle;
  * every such ray's proxy is reached by the sphere-against-cone test of the binning (cone_hit).

The device code evaluates the same expressions in fp32 with explicit safety margins; its end-to-end check is the bitwise equality of the
list path and the tree walk (tests/test_grt_gpu.py).  This file guards the formulas themselves."""
import numpy as np


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _cone_rays(rng, axis, theta, n):
    """n unit directions within `theta` of `axis` (some on the rim)."""
    a = axis / np.linalg.norm(axis)
    u = np.cross(a, [1.0, 0.0, 0.0] if abs(a[0]) < 0.9 else [0.0, 1.0, 0.0])
    u /= np.linalg.norm(u)
    v = np.cross(a, u)
    ang = theta * np.sqrt(rng.uniform(0, 1, n))
    ang[: n // 4] = theta
    phi = rng.uniform(0, 2 * np.pi, n)
    return (np.cos(ang)[:, None] * a + np.sin(ang)[:, None] * (np.cos(phi)[:, None] * u + np.sin(phi)[:, None] * v))


def _packet_bounds(axis, theta, v, W, k, dmin, dmax):
    """packet_bounds of grt_kernels.hip without its rounding margins."""
    L = np.linalg.norm(v)
    chord = np.sqrt(max(0.0, 2 * (1 - np.cos(theta))))
    w = W @ axis                      # (S^-1 e)_i = W_i . dh
    s = k * k * w                     # (S e)_i = kscl_i^2 (W_i . dh)
    se = np.linalg.norm(s) + k.max() * chord
    sie = np.linalg.norm(w) + chord / k.min()
    h = np.sqrt(3.0) * np.sqrt(max(0.0, se * se - 1.0 / (sie * sie)))
    ca = np.clip(v @ axis / L, -1, 1) if L > 0 else 1.0
    sa = np.sqrt(max(0.0, 1 - ca * ca))
    ct, st = np.cos(theta), np.sin(theta)
    cmax = 1.0 if ca >= ct else min(1.0, ca * ct + sa * st)
    cmin = -1.0 if ca <= -ct else max(-1.0, ca * ct - sa * st)
    tl, th = L * cmin - h, L * cmax + h
    lo = tl / dmax if tl > 0 else 0.0
    hi = th / dmin if th > 0 else 0.0
    return lo, hi


def test_hit_distance_bounds_and_cone_test_hold():
    rng = np.random.default_rng(11)
    checked = 0
    for case in range(600):
        k = np.exp(rng.normal(size=3) * rng.choice([0.2, 0.8, 1.6])) * rng.choice([0.01, 0.05, 0.3])      # kscl, anisotropy up to ~100
        R = _rot(rng)
        W = (R / k).T                                                                                    # W = diag(1/kscl) R^T
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        theta = rng.choice([0.002, 0.01, 0.05, 0.3])
        # centre somewhere near the cone (in front, beside, or around the apex)
        dist = rng.choice([0.0, 0.02, 0.5, 3.0, 30.0])
        off = rng.normal(size=3) * rng.choice([0.0, 0.5, 2.0]) * max(k.max(), dist * np.tan(theta))
        mu = axis * dist + off
        o = np.zeros(3)
        v = mu - o
        dirs = _cone_rays(rng, axis, theta, 400)
        lens = rng.uniform(0.5, 2.0, size=len(dirs))
        dmin, dmax = lens.min(), lens.max()
        d = dirs * lens[:, None]
        po = W @ (o - mu)
        pd = d @ W.T
        t = -(pd @ po) / (pd * pd).sum(1)
        # unit-box slab test in the proxy's frame
        with np.errstate(divide="ignore", invalid="ignore"):
            a0, a1 = (-1 - po) / pd, (1 - po) / pd
        tn, tf = np.minimum(a0, a1).max(1), np.maximum(a0, a1).min(1)
        touch = (tn <= tf) & (t > 0)
        if not touch.any():
            continue
        checked += int(touch.sum())
        Rs, Rt, L = np.linalg.norm(k), np.sqrt(3.0) * k.max(), np.linalg.norm(v)
        key = max(0.0, (L - Rt) / dmax)
        ub = (L + Rt) / dmin
        lo, hi = _packet_bounds(axis, theta, v, W, k, dmin, dmax)
        tt = t[touch]
        assert (tt >= key - 1e-9 * (1 + L)).all() and (tt <= ub + 1e-9 * (1 + L)).all(), case
        assert (tt >= lo - 1e-9 * (1 + L)).all() and (tt <= hi + 1e-9 * (1 + L)).all(), (case, tt.min(), tt.max(), lo, hi)
        # cone_hit: distance from the sphere's centre to the cone's surface <= s cos - c sin
        c = v @ axis
        sq = np.sqrt(max(0.0, L * L - c * c))
        assert sq * np.cos(theta) - c * np.sin(theta) <= Rs * (1 + 1e-9) + 1e-12, case
    assert checked > 20000


def test_bounds_are_exact_for_a_sphere_seen_along_the_axis():
    """Isotropic proxy, zero-width cone: the interval collapses onto the hit distance itself (|mu - o| for the ray through the centre)."""
    k = np.full(3, 0.07)
    W = np.eye(3) / k
    axis = np.array([0.0, 0.0, 1.0])
    v = np.array([0.0, 0.0, 5.0])
    lo, hi = _packet_bounds(axis, 0.0, v, W, k, 1.0, 1.0)
    assert abs(lo - 5.0) < 1e-9 and abs(hi - 5.0) < 1e-9
