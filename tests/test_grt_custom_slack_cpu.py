"""CPU: the pruning slack of the custom primitives (csrc/grt_kernels.hip: grt_proxy_kernel) is conservative - as mathematics, in float64.

render.primitive_type = custom offers a particle to every ray that touches its WORLD box and reports the distance t* of maximum response,
the minimiser of |W (o + t d - mu)| with W = diag(1 / kscl) R^T (intersectCustomParticle, gaussianParticles.cuh:407-441).  The tree walk
prunes a node when `box entry - slack > k-th best distance so far`: sound only if t* >= entry - slack for every ray through the box.  The
bounds in use: slack = kmax |s|, s_i = sum_j |W_ij| h_j (h = the box's half extents) for every ray, and 3 max scl + half diagonal for the
candidates the program can accept (the kernel takes the smaller); the Euclidean half diagonal that preceded them fails
for anisotropic particles - shown here too, so that the test would have caught the defect the 1 M-particle parity run found in round 5."""
import numpy as np


def _rot(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_custom_primitive_slack_bounds_the_reported_distance_for_every_ray_through_the_world_box():
    rng = np.random.default_rng(11)
    worst_new, worst_old, worst_acc, n_rays, n_accepted = 0.0, 0.0, 0.0, 0, 0
    for _ in range(400):
        R = _rot(rng.normal(size=4))
        k = np.exp(rng.uniform(np.log(0.01), np.log(1.0), size=3))          # kscl: anisotropy up to 100
        mu = rng.normal(size=3)
        W = np.diag(1.0 / k) @ R.T
        h = np.abs(R) @ k                                                   # half extents of the world box of the oriented box (particlePrimitives.cu:498-541)
        slack_new = k.max() * np.linalg.norm(np.abs(W) @ h)
        slack_old = np.linalg.norm(h)
        for _ in range(200):
            p = mu + rng.uniform(-1, 1, size=3) * h                         # a point of the box the ray passes through
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            o = p - rng.uniform(0.5, 5.0) * d
            j = 1.0 / d
            a0, a1 = (mu - h - o) * j, (mu + h - o) * j
            tnear, tfar = np.minimum(a0, a1).max(), np.maximum(a0, a1).min()
            assert tnear <= tfar + 1e-12
            po, pd = W @ (o - mu), W @ d
            t_star = -(po @ pd) / (pd @ pd)
            worst_new = max(worst_new, (tnear - t_star) / slack_new, (t_star - tfar) / slack_new)
            worst_old = max(worst_old, (tnear - t_star) / slack_old)
            # the second bound holds for the candidates the program can ACCEPT (|W (x* - mu)| ks < 3; the kernel uses the smaller of the two)
            for ks in (1.5, 3.0, 4.5):                                       # kernelScale of dense ... faint particles (kscl = ks * scl)
                slack_acc = 3.0 * k.max() / ks + np.linalg.norm(h)
                if np.linalg.norm(po + t_star * pd) * ks < 3.0:
                    worst_acc = max(worst_acc, (tnear - t_star) / slack_acc, (t_star - tfar) / slack_acc)
                    n_accepted += 1
            n_rays += 1
    assert worst_new <= 1.0, f"t* precedes the box entry by {worst_new:.3f} x the slack"
    assert worst_acc <= 1.0 and n_accepted > 5000, (worst_acc, n_accepted)
    assert worst_old > 1.5, "the Euclidean half diagonal was expected to fail on anisotropic particles"   # (the defect this test guards against)
    assert n_rays == 80000
