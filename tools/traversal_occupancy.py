#!/usr/bin/env python
"""Throughput of the traversal-only kernel (k_intersect_batch, the wide walk) on incoherent rays:
what a split (extend / shade) design could get out of more resident wavefronts.  Development
builds with -DYT_IB_WAVES=N bound the kernel to N waves per SIMD.
  SCENE=cornell1m N=4000000 YTHIP_LIB=build/dev/libythip_ibN.so python tools/traversal_occupancy.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ythip as yt, scenes as ysc
SCENE = os.environ.get("SCENE", "cornell1m")
if SCENE == "cornell1m":
    import parity as P
    flat = P.scene_cornell_1m()
elif SCENE == "cfg4":
    flat = ysc.instanced_scene()
elif SCENE == "cfg5":
    flat = ysc.hair_scene()
else:
    flat = ysc.plane_scene()
n = int(os.environ.get("N", "4000000"))
rng = np.random.default_rng(7)
lo, hi = flat.positions.min(0), flat.positions.max(0)
rays = np.zeros(n, yt.ray_dt)
rays["o"] = (lo + (hi - lo) * (0.05 + 0.9 * rng.random((n, 3)))).astype("f4")
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays["d"] = d.astype("f4"); rays["tmin"] = 1e-4; rays["tmax"] = np.finfo("f4").max
ctx = yt.Context(0)
ctx.upload_scene(flat); ctx.make_trace_bvh(flat); ctx.make_trace_lights(flat)
h = ctx.intersect_batch(rays)
ctx.set_profiling(1); ctx.reset_stats()
for _ in range(3):
    h = ctx.intersect_batch(rays)
s = ctx.get_stats(); ctx.set_profiling(0)
ms = s["trace_ms"] / s["trace_launches"]
import hashlib
print(f"{SCENE} {n} incoherent rays: {ms:8.3f} ms  {n / ms / 1e3:8.1f} Mrays/s  hits {int(h['hit'].sum())}  digest {hashlib.sha1(h.tobytes()).hexdigest()[:12]}")
