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


def test_tangent_plane_rectangles_select_every_packet_a_ray_can_hit_from():
    """The candidate selection of the binning (grt_kernels.hip: grid_particle, rect_overlap; DESIGN.md §5 "late round 3"): in a frame
    (u, w, a) with d.a > 0 for every ray, a ray that touches the proxy box has its tangent-plane coordinates ((d.u)/(d.a), (d.w)/(d.a))
    inside the bounding rectangle of the box's 8 projected corners whenever the whole box lies in front of the apex plane — so a packet
    whose own rectangle of ray coordinates misses that rectangle holds no such ray.  Checked in float64 against brute force; boxes that
    come near the plane (the device walks the super tiles for those) are checked to be recognised by the zmin test."""
    rng = np.random.default_rng(5)
    checked = rejected = 0
    for case in range(500):
        k = np.exp(rng.normal(size=3) * rng.choice([0.2, 0.8, 1.6])) * rng.choice([0.01, 0.05, 0.3])
        R = _rot(rng)
        W = (R / k).T
        a = _rot(rng)[:, 2]
        u = np.cross(a, [1.0, 0.0, 0.0] if abs(a[0]) < 0.9 else [0.0, 1.0, 0.0])
        u /= np.linalg.norm(u)
        w = np.cross(a, u)
        # a proxy somewhere in front of (or beside) the apex, seen under up to ~85 degrees off the axis
        dirc = _cone_rays(rng, a, np.deg2rad(rng.choice([20, 60, 85])), 1)[0]
        v = dirc * rng.uniform(0.05, 3.0)
        H = R * k                                       # columns = the box's half axes in world space
        corners = np.array([v + H @ np.array([s0, s1, s2]) for s0 in (-1, 1) for s1 in (-1, 1) for s2 in (-1, 1)])
        z = corners @ a
        ext = np.linalg.norm(v) + np.linalg.norm(k)
        if z.min() < 0.02 * ext:
            rejected += 1                               # no rectangle: the walk through the super tiles serves this box
            continue
        x0, x1 = (corners @ u / z).min(), (corners @ u / z).max()
        y0, y1 = (corners @ w / z).min(), (corners @ w / z).max()
        # rays through random points of the box (all of them touch it), and rays aimed around it
        pts = v + (H @ rng.uniform(-1, 1, size=(3, 400))).T
        pts = np.concatenate([pts, v + (H @ (rng.uniform(-1, 1, size=(3, 400)) * 1.5)).T])
        d = pts / np.linalg.norm(pts, axis=1, keepdims=True)
        d = d[d @ a > 0.05]
        po, pd = W @ (-v), (W @ d.T).T                  # the ray in the proxy's frame: origin W (o - mu), direction W d
        with np.errstate(divide="ignore", invalid="ignore"):
            t0, t1 = (-1 - po) / pd, (1 - po) / pd
        tn, tf = np.minimum(t0, t1).max(axis=1), np.maximum(t0, t1).min(axis=1)
        touch = (tn <= tf) & (tf > 0)
        tx, ty = d @ u / (d @ a), d @ w / (d @ a)
        inside = (tx >= x0 - 1e-9) & (tx <= x1 + 1e-9) & (ty >= y0 - 1e-9) & (ty <= y1 + 1e-9)
        assert inside[touch].all(), f"case {case}: a ray touching the box lies outside the projected rectangle"
        checked += int(touch.sum())
    assert checked > 50_000 and rejected > 20


def test_refined_intervals_cover_every_ray_that_can_want_the_entry():
    """list_round<REFINE> replaces an entry's geometric hit-distance interval by [min, max] of t over the rays that were running AND met
    the proxy at the packet's first test of the entry.  The argument that this loses nothing, as a simulation: rays advance through a
    shared list in rounds of k candidates, stop at random, and an entry is only offered while its interval overlaps the union of the
    running rays' windows — every ray must still receive exactly the candidates a brute-force per-ray scan gives it."""
    rng = np.random.default_rng(23)
    for case in range(60):
        n_rays, n_ent, k = 16, int(rng.integers(20, 200)), 4
        t = rng.uniform(0, 10, size=(n_ent, n_rays))                    # hit distance of entry e for ray r
        meets = rng.uniform(size=(n_ent, n_rays)) < rng.choice([0.1, 0.5, 0.9])
        stop_after = rng.integers(1, 40, size=n_rays)                   # a ray terminates after this many processed hits
        geo_lo, geo_hi = t.min(axis=1) - rng.uniform(0, 3, n_ent), t.max(axis=1) + rng.uniform(0, 3, n_ent)
        want = []
        for r in range(n_rays):                                         # brute force: the ray's hits in order, cut at its termination
            order = [e for e in np.argsort(t[:, r], kind="stable") if meets[e, r]]
            want.append(order[: stop_after[r]])
        lo, hi = geo_lo.copy(), geo_hi.copy()
        refined = np.zeros(n_ent, bool)
        tmin = np.full(n_rays, -1.0)
        running = np.ones(n_rays, bool)
        got = [[] for _ in range(n_rays)]
        while running.any():
            wmin = tmin[running].min()
            bufs = [[] for _ in range(n_rays)]
            bound = np.full(n_rays, np.inf)
            for e in range(n_ent):                                      # the scan: entries whose interval overlaps the wave's window
                wmax = bound[running].max()
                if hi[e] < wmin or lo[e] > wmax:
                    continue
                if not refined[e]:
                    m = running & meets[e]
                    lo[e], hi[e] = (t[e, m].min(), t[e, m].max()) if m.any() else (np.inf, -np.inf)
                    refined[e] = True
                for r in np.nonzero(running & meets[e])[0]:
                    if tmin[r] < t[e, r] < bound[r]:
                        bufs[r] = sorted(bufs[r] + [(t[e, r], e)])[:k]
                        if len(bufs[r]) == k:
                            bound[r] = bufs[r][-1][0]
            for r in np.nonzero(running)[0]:
                if not bufs[r]:
                    running[r] = False
                    continue
                for tt, e in bufs[r]:
                    if len(got[r]) < stop_after[r]:
                        got[r].append(e)
                        tmin[r] = tt
                if len(got[r]) >= stop_after[r]:
                    running[r] = False
        for r in range(n_rays):
            assert got[r] == want[r], f"case {case}, ray {r}"


def test_ordered_hit_words_sort_like_distance_particle_pairs():
    """HitBufferT (grt_kernels.hip) keeps a candidate as ONE double: the float distance widened (exact) with the particle index in the
    29 low mantissa bits the widening leaves zero, and inserts with a[k] = max(a[k-1], min(a[k], K)).  Checked here: the widening leaves
    those bits zero for every positive float (normal or denormal), the words order exactly like (distance, particle) pairs do — ties in
    the distance included — and the min / max recurrence IS the sorted insertion that drops the largest."""
    rng = np.random.default_rng(2)
    t = np.concatenate([rng.uniform(1e-3, 50, 4000), np.exp(rng.uniform(-100, 80, 4000)), [1e-45, 1.4e-45, 3e-39, 3.0e38, 1.0, 1.0, 1.0]]).astype(np.float32)
    t = np.concatenate([t, rng.choice(t, 3000)])                     # plenty of equal distances
    ids = rng.integers(0, 2**29 - 1, size=t.size, dtype=np.uint64)
    wide = t.astype(np.float64).view(np.uint64)
    assert not np.any(wide & np.uint64(0x1FFFFFFF)), "a widened float has bits in the index field"
    words = (wide | ids).view(np.float64)
    order_words = np.argsort(words, kind="stable")
    order_pairs = np.lexsort((ids, t))
    assert np.array_equal(words[order_words], words[order_pairs])
    assert np.array_equal(t[order_words], t[order_pairs]) and np.array_equal(ids[order_words], ids[order_pairs])
    # decoding gives the pair back
    back_t = (words.view(np.uint64) & ~np.uint64(0x1FFFFFFF)).view(np.float64).astype(np.float32)
    assert np.array_equal(back_t, t) and np.array_equal(words.view(np.uint64) & np.uint64(0x1FFFFFFF), ids)
    # the recurrence against a sort, 16 slots, a stream of 300 candidates
    empty = (np.float32(3.0e38).astype(np.float64).view(np.uint64) | np.uint64(0x1FFFFFFF)).view(np.float64)
    a = np.full(16, empty)
    seen = []
    for K in words[rng.permutation(words.size)[:300]]:
        if K >= empty:
            continue
        seen.append(K)
        prev = a.copy()
        for k in range(15, 0, -1):
            a[k] = max(prev[k - 1], min(prev[k], K))
        a[0] = min(prev[0], K)
        want = np.sort(np.array(seen))[:16]
        assert np.array_equal(a[:want.size], want)
