#!/usr/bin/env python
"""Single-GPU probe of the N-GPU strong-scaling bound on any workload: time the slice rank 0 of N would
render (16-pixel tile columns dealt round-robin, no gather).  SCENE=plane|cfg4|cornell1m RES= SPP="""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ythip as yt, scenes as ysc
SCENE = os.environ.get("SCENE", "plane")
if SCENE == "cfg4":
    flat, res, spp = ysc.instanced_scene(), 1920, 256
elif SCENE == "cornell1m":
    import parity as P
    flat, res, spp = P.scene_cornell_1m(), 1024, 64
else:
    flat, res, spp = ysc.plane_scene(), 1280, 64
res, spp = int(os.environ.get("RES", res)), int(os.environ.get("SPP", spp))
ctx = yt.Context(0)
ctx.upload_scene(flat); ctx.make_trace_bvh(flat); ctx.make_trace_lights(flat)
base = None
for n in [1, 2, 4, 8]:
    p = yt.trace_params(sampler="path", resolution=res, samples=1 << 30, batch=spp)
    ctx.make_trace_state(flat, p, cols=(0, n))
    ctx.trace_samples(p); ctx.trace_samples(p)
    ctx.set_profiling(1); ctx.reset_stats()
    for _ in range(3):
        ctx.trace_samples(p)
    s = ctx.get_stats(); ctx.set_profiling(0)
    ms = s["trace_ms"] / s["trace_launches"]
    base = base or ms
    print(f"{SCENE} {res}x{spp}spp  N={n}  {ms:9.3f} ms  {ctx.npixels * spp / ms / 1e3:9.1f} Msamples/s of the slice   speed-up {base / ms:5.2f}x")
