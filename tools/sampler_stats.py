#!/usr/bin/env python
"""Work counters per sample (the counting launch: binary walk, the reference's counts) for
several samplers on one scene:  SCENE=materials SAMPLERS=path,pathdirect,pathmis python tools/sampler_stats.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ythip as yt
import parity as P
flat = P.SCENES[os.environ.get("SCENE", "materials")]()
ctx = P.gpu_context(flat)
res, spp = int(os.environ.get("RES", "360")), int(os.environ.get("SPP", "8"))
for sampler in os.environ.get("SAMPLERS", "path,pathdirect,pathmis").split(","):
    p = yt.trace_params(sampler=sampler, resolution=res, samples=spp, batch=spp)
    ctx.make_trace_state(flat, p)
    ctx.set_profiling(2); ctx.reset_stats()
    ctx.trace_samples(p); ctx.sync()
    s = ctx.get_stats(); ctx.set_profiling(0)
    n = max(s["samples"], 1)
    print(sampler, {k: round(v / n, 2) for k, v in s.items() if k in ("rays", "nodes", "triangles", "quads", "lines", "points", "instances", "shades")}, "samples", s["samples"])
