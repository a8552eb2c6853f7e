"""Where do the own walk's hit records differ from the exact walk's?  SCENE=cfg5 python tools/r05/own_diag.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import parity as P
from parity import yt, ysc
name = os.environ.get("SCENE", "cfg5")
flat = {"cfg5": ysc.hair_scene, "cfg2": ysc.plane_scene, "cfg4": ysc.instanced_scene, "cfg2b": P.scene_cornell_1m}.get(name, P.SCENES.get(name))()
ctx = P.gpu_context(flat); ctx.make_own_bvh(flat)
p = yt.trace_params(resolution=1280, samples=1); ctx.make_trace_state(flat, p)
cam = ctx.camera_rays(p); rnd = P.random_rays(flat, 200_000, seed=23)
for label, rays in (("camera rays", cam), ("random rays", rnd)):
    ex, ow = ctx.intersect_batch(rays), ctx.intersect_batch_own(rays)
    both = (ex["hit"] != 0) & (ow["hit"] != 0)
    same = both & (ex["instance"] == ow["instance"]) & (ex["element"] == ow["element"])
    diff = both & ~same
    flip_a, flip_b = (ex["hit"] != 0) & (ow["hit"] == 0), (ex["hit"] == 0) & (ow["hit"] != 0)
    d = ow["distance"].astype(np.float64) - ex["distance"].astype(np.float64)
    rel = np.abs(d) / np.maximum(np.abs(ex["distance"]), 1e-6)
    print(f"{label}: {len(rays)} rays, both hit {both.sum()}, same primitive {same.sum()}, other primitive {diff.sum()} "
          f"(own nearer {int((diff & (d < 0)).sum())}, own farther {int((diff & (d > 0)).sum())}, equal {int((diff & (d == 0)).sum())}), "
          f"exact-only hits {flip_a.sum()}, own-only hits {flip_b.sum()}")
    if diff.any():
        q = np.quantile(rel[diff], [0.1, 0.5, 0.9, 0.99])
        print("   relative distance difference of the other-primitive cases: p10 %.2e p50 %.2e p90 %.2e p99 %.2e" % tuple(q))
        inst_ex = np.bincount(ex["instance"][diff], minlength=len(flat.instances)); inst_ow = np.bincount(ow["instance"][diff], minlength=len(flat.instances))
        print("   instances named (exact / own):", inst_ex[:8], inst_ow[:8])
        k = np.flatnonzero(diff)[:5]
        for i in k:
            print("   e.g.", rays[i], "\n        exact", ex[i], "\n        own  ", ow[i])
    if same.any():
        print("   same primitive: max rel distance difference %.2e, max |du| %.2e" % (rel[same].max(), np.abs(ow["u"][same] - ex["u"][same]).max()))
ctx.close()
