"""In-tree build of libgrut_amd.so with hipcc for gfx950 (cross-compiles without a GPU).

    python 3dgrut_amd/build.py [--force] [--verbose]

One object per .hip file, rebuilt only when its sources are newer; linked into csrc/libgrut_amd.so,
which travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libgrut_amd.so")
SOURCES = ["scan_sort.hip", "gut_kernels.hip", "gut_poses.hip", "gut_render.hip", "gut_api.hip", "grt_kernels.hip", "grt_api.hip", "optim.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-ffp-contract=fast",
         # SLP-packing scalar f32 math into v_pk_*_f32 costs register-pair shuffles (v_mov) and VGPRs on gfx950
         "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
# per-file additions.  grt_kernels.hip evaluates hit distances under `#pragma clang fp contract(off)` so that the per-ray
# hit ORDER is reproducible bit for bit by the CPU checker; the pragma is only honoured when the command-line mode is
# `on` (with `fast` the backend fuses regardless), so that file is built with -ffp-contract=on.
# gut_poses.hip derives the frame's poses the way the reference's HOST code does (numpy / torch / host C++: nothing contracts there).
# gut_render.hip (the compositing sweeps: fp32 VALU issue while their waves run, latency in the launch's ramp-down): the backend's max-ILP
# scheduling strategy instead of the default occupancy-first one - the kernels' occupancy is pinned by amdgpu_waves_per_eu anyway.  Same
# instructions in another order: gradient sweep 0.807 -> 0.792 ms, forward 0.449 -> 0.443 ms, step -1.1 % (A/B on one box, interleaved:
# scripts/ab_sched_variants.sh, profiles/r06_sched_strategy_ab.txt).  Measured and NOT applied: gut_kernels.hip (the gradient gather loses
# 12 us), grt_kernels.hip (trace forward -0.1 ms, replay backward +0.03: one file), max-memory-clause (slower everywhere).
# The sorted hit buffer's kernels (SH radiance) LOSE with it (K = 16 frame 10.87 -> 11.32 ms), so gut_render.hip is compiled twice: part 0
# (everything else, max-ILP) and part 1 (launch_render_k_fwd / _bwd and the kernels they instantiate, default strategy) - see the file's head.
FILE_FLAGS = {"grt_kernels.hip": ["-ffp-contract=on"], "gut_poses.hip": ["-ffp-contract=off"]}
# translation units: (source, object, flags of this unit)
UNITS = [(s_, s_.replace(".hip", ".o"), []) for s_ in SOURCES if s_ != "gut_render.hip"] + [
    ("gut_render.hip", "gut_render.o", ["-DGRUT_RENDER_PART=0", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
    ("gut_render.hip", "gut_render_k.o", ["-DGRUT_RENDER_PART=1"])]


def unit_flags(src: str):
    """Per-file + per-unit flags of `src`'s FIRST translation unit (what scripts/kernel_resources.py and valu_model.py compile with)."""
    return [*FILE_FLAGS.get(src, []), *next((f for s_, _, f in UNITS if s_ == src), [])]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h", ".inl"))] + \
           [os.path.join(HERE, "..", "include", "grut_amd.h")]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    hipcc = _hipcc()
    deps = _deps()
    objs, jobs = [], []
    for src, obj, unit_flags in UNITS:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = os.path.join(CSRC, obj)
        objs.append(op)
        if force or _stale(op, [sp] + deps):
            jobs.append([hipcc, "-x", "hip", *FLAGS, *FILE_FLAGS.get(src, []), *unit_flags, *extra_flags, "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr:
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(OUT, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
