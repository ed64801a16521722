import importlib, sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
import argparse
a = argparse.Namespace(steps=3, warmup=2)
for env in ({}, {"GRUT_GRT_NO_LISTS": "1"}):
    for k in ("GRUT_GRT_NO_LISTS",): os.environ.pop(k, None)
    os.environ.update(env)
    # patch: capture tracer stats through a global hook
    pt = importlib.import_module("3dgrut_amd.playground_tracer")
    orig = pt.Tracer.render_playground
    holder = {}
    def wrapped(self, *args, **kw):
        holder["tr"] = self
        return orig(self, *args, **kw)
    pt.Tracer.render_playground = wrapped
    r = bench.bench_hybrid(a, "cuda", 2_000_000, 1920, 1080, 0.008, emit=False)
    pt.Tracer.render_playground = orig
    st = holder["tr"].tracer_wrapper.stats()
    print(env, "ms", r["ms_per_step"], "list_entries", int(st.list_entries), "mirror", r["rays_with_mirror_bounce"], "opacity", r["mean_opacity"], flush=True)
