#!/usr/bin/env python
"""Time ythip_denoise_state on a resident 1920x1080 frame (hipEvents through torch are not
needed: the call synchronises; wall time of repeated calls).  python tools/denoise_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd"))
import ythip as yt, scenes as ysc
flat = ysc.instanced_scene()
ctx = yt.Context(0)
ctx.upload_scene(flat); ctx.make_trace_bvh(flat); ctx.make_trace_lights(flat)
p = yt.trace_params(sampler="path", resolution=1920, samples=4, batch=4)
ctx.make_trace_state(flat, p)
ctx.trace_samples(p)
for levels in (5, 3):
    ctx.denoise_state(download=False, levels=levels)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        ctx.denoise_state(download=False, levels=levels)
    dt = (time.perf_counter() - t0) / n
    px = ctx.width * ctx.height
    alg = px * (60 + levels * 64 + 32 + 16)   # prep 40 in / 48 out... see DESIGN §7f
    print(f"denoise {ctx.width}x{ctx.height} levels={levels}: {dt*1e3:.3f} ms  ({px/dt/1e9:.2f} Gpixel/s, "
          f"{px*(levels*64+128)/dt/1e9:.0f} GB/s algorithmic)")
