"""The half-tile culling of the 3DGUT forward (3dgrut_amd/csrc/gut_render.hip: WavePyramid, pyramid_misses),
checked as mathematics in float64 against brute force.  The kernel drops a staged entry when, in the particle's canonical frame, the sphere of
radius r = sqrt(gmax) around the origin lies wholly in front of the image of the camera plane AND wholly outside one face of the pyramid
that bounds the wave's rays.  The accept test it must never pre-empt is the LINE - sphere test  |v x u|^2 < gmax |v|^2  (u = M (o - mu),
v = M d): here, for random anisotropic particles, random ray bundles and random sphere sizes, every dropped entry is rejected by every ray
of the bundle (rays sampled densely inside the bundle's bounding rectangle, corners and edges included), and the test is not vacuous (it
drops a good share of the entries that no ray accepts)."""
import numpy as np


def _frame(a):
    k = np.eye(3)[np.argmin(np.abs(a))]
    e1 = np.cross(a, k)
    e1 /= np.linalg.norm(e1)
    return e1, np.cross(a, e1)


def _pyramid(dirs):
    """wave_pyramid without its rounding pads: axis = first ray, tangent rectangle of all rays."""
    a = dirs[0] / np.linalg.norm(dirs[0])
    e1, e2 = _frame(a)
    q = dirs @ a
    assert (q > 0.5 * np.linalg.norm(dirs, axis=1)).all()
    x, y = (dirs @ e1) / q, (dirs @ e2) / q
    return a + x.min() * e1 + y.min() * e2, e1, e2, x.max() - x.min(), y.max() - y.min()


def _misses(M, u, r, c00, e1, e2, dx, dy):
    """pyramid_misses without its rounding slack."""
    v00, f1, f2 = M @ c00, M @ e1, M @ e2
    v10, v01 = v00 + f1 * dx, v00 + f2 * dy
    nx0, nx1, ny0, ny1, wf = np.cross(v00, f2), np.cross(v10, f2), np.cross(v00, f1), np.cross(v01, f1), np.cross(f1, f2)
    sx0, sx1, sy0, sy1, sf = -u @ nx0, u @ nx1, u @ ny0, -u @ ny1, -u @ wf
    front = sf > r * np.linalg.norm(wf)
    out = (sx0 > r * np.linalg.norm(nx0)) or (sx1 > r * np.linalg.norm(nx1)) or (sy0 > r * np.linalg.norm(ny0)) or (sy1 > r * np.linalg.norm(ny1))
    return bool(front and out)


def _rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_half_tile_cull_never_drops_an_entry_a_ray_of_the_wave_accepts():
    rng = np.random.default_rng(20)
    dropped = unaccepted = total = 0
    for trial in range(4000):
        # a bundle of rays around a random axis: 16 x 8 "pixels" of 1e-3 rad, through a mildly non-linear camera (second-order terms)
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        t1, t2 = _frame(axis)
        gx, gy = np.meshgrid(np.linspace(-8, 8, 16) * 1e-3, np.linspace(-4, 4, 8) * 1e-3)
        warp = 1.0 + rng.uniform(-30, 30) * (gx ** 2 + gy ** 2)
        dirs = axis[None] + (gx * warp).reshape(-1, 1) * t1[None] + (gy * warp).reshape(-1, 1) * t2[None]
        dirs *= rng.uniform(0.5, 2.0, size=(len(dirs), 1))   # (the test is scale-invariant in the direction)
        origin = rng.normal(size=3)
        # a particle somewhere around the bundle, in front or behind, needle- to pancake-shaped
        scale = np.exp(rng.uniform(np.log(2e-3), np.log(0.5), size=3))
        M = np.diag(1.0 / scale) @ _rotation(rng).T
        depth = rng.uniform(-1.0, 6.0)
        mu = origin + depth * axis + rng.normal(size=3) * abs(depth) * rng.choice([2e-3, 1e-2, 5e-2])
        r = np.sqrt(rng.uniform(0.5, 12.0))
        u = M @ (origin - mu)
        c00, e1, e2, dx, dy = _pyramid(dirs)
        cull = _misses(M, u, r, c00, e1, e2, dx, dy)
        # brute force: the bundle's rays and a dense resampling of its tangent rectangle (every line the pyramid bounds)
        a = dirs[0] / np.linalg.norm(dirs[0])
        s, tt = np.meshgrid(np.linspace(0, 1, 33), np.linspace(0, 1, 33))
        dense = c00[None] + (s.reshape(-1, 1) * dx) * e1[None] + (tt.reshape(-1, 1) * dy) * e2[None]
        allv = np.concatenate([dirs, dense]) @ M.T
        cc = (np.cross(allv, u[None]) ** 2).sum(1)
        accepted = bool((cc < r * r * (allv ** 2).sum(1)).any())
        total += 1
        unaccepted += not accepted
        if cull:
            dropped += 1
            assert not accepted, f"trial {trial}: a dropped entry is accepted by a ray of the bundle"
    assert unaccepted > 500 and dropped > 0.5 * unaccepted, (total, unaccepted, dropped)
