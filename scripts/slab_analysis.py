"""How much of the binning work the compositing ever reads (CPU, through the oracle): DESIGN.md §7b "entries nobody reads".

For the bench workload (1 M Gaussians, 1080p) the oracle reports, per pixel, how many entries of its tile list the forward loop
examines before the ray ends.  A tile is done when all of its pixels are; entries behind that point are expanded, sorted and
range-scanned for nothing.  If the list were built in S depth slabs (particles are depth-sorted before the expansion), each slab
expanded only into tiles still alive after the previous one, the number of entries built would be the figure printed per S."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

syn = importlib.import_module("workloads.synthetic")
camera = importlib.import_module("3dgrut_amd.camera")
n, W, H = (int(a) for a in (sys.argv[1:4] + ["1000000", "1920", "1080"][len(sys.argv) - 1:]))
F = np.float32
d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=0.01)
K = syn.pinhole_intrinsics(W, H)
ro, rd = syn.pinhole_rays(W, H, K)
cam, ps, pe = camera.camera_from_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(0, n_views=8)[None], intrinsics=K))
cfg = oracle.default_gut_config()
t = time.time()
proj = oracle.gut_project(cfg, cam, ps, pe, 3, d12, sph)
bins = oracle.gut_bin(cfg, W, H, proj)
I = bins["num_intersections"]
print(f"N {n}, {W}x{H}: I = {I} ({time.time() - t:.1f} s)")
p = lambda a: a.ctypes.data_as(C.c_void_p)
consumed = np.zeros(W * H, np.uint32)
ps_, pe_ = np.asarray(ps, F), np.asarray(pe, F)
ro_, rd_ = np.ascontiguousarray(ro, F).reshape(H, W, 3), np.ascontiguousarray(rd, F).reshape(H, W, 3)
t = time.time()
oracle.lib(F).orc_gut_render_fwd_consumed(C.byref(cfg), W, H, p(ps_), p(pe_), p(np.ascontiguousarray(d12, F)), p(bins["sorted_idx"]),
                                          p(bins["tile_ranges"]), p(ro_), p(rd_), p(consumed))
print(f"forward loop: {time.time() - t:.1f} s; entries examined per pixel: mean {consumed.mean():.1f}")
gx, gy = (W + 15) // 16, (H + 15) // 16
cons = np.zeros((gy * 16, gx * 16), np.uint32)
cons[:H, :W] = consumed.reshape(H, W)
need = cons.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(gy * gx, 256).max(1).astype(np.int64)   # entries a tile reads
rng = bins["tile_ranges"].astype(np.int64)
lens = rng[:, 1] - rng[:, 0]
print(f"entries read by some pixel of their tile: {need.sum()} of {lens.sum()} ({100.0 * need.sum() / lens.sum():.1f} %)")
# depth slabs by particle rank: rank of a particle = position in the (depth bits, index) order of the visible particles
depth_bits = proj["depth"].view(np.uint32).astype(np.uint64)
order = np.argsort((depth_bits << np.uint64(32)) | np.arange(n, dtype=np.uint64), kind="stable")
rank = np.empty(n, np.int64)
rank[order] = np.arange(n)
entry_rank = rank[bins["sorted_idx"].astype(np.int64)]
tile_of = np.repeat(np.arange(gx * gy), lens)
pos_in_tile = np.arange(I) - rng[tile_of, 0]
for S in (2, 4, 8, 16, 32):
    slab = entry_rank * S // n
    # a tile is alive at the start of slab s iff it still reads an entry of slab >= s: its first slab-s entry lies before need[tile]
    first_of_slab = np.ones(I, bool)
    first_of_slab[1:] = (slab[1:] != slab[:-1]) | (tile_of[1:] != tile_of[:-1])
    alive_slab = np.zeros(I, bool)
    starts = np.nonzero(first_of_slab)[0]
    alive = pos_in_tile[starts] < need[tile_of[starts]]
    run_len = np.diff(np.append(starts, I))
    built = int(run_len[alive].sum())
    print(f"S = {S:2d} slabs: {built} entries built ({100.0 * built / I:.1f} % of I)")
