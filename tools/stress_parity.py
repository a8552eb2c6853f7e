#!/usr/bin/env python
"""One-off stress of the hit-record parity (GPU wide walk vs the live reference):
millions of seeded rays per scene incl. axis-aligned / zero-component directions,
finite tmax, rays starting on surfaces.  Diagnostic; the committed tests use
100k-ray samples of the same generators."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import parity as P
from parity import yt, ry

N = int(os.environ.get("NRAYS", "2000000"))
names = list(P.SCENES) + [n for n in P.ref_scene_names() if n in ("features1", "shapes1", "instances1", "arealights1")]
bad_total = 0
for name in names:
    flat = P.SCENES[name]() if name in P.SCENES else P.load_ref_scene(name)
    ctx = P.gpu_context(flat); rb = P.RefBundle(flat)
    for seed in (101, 202):
        rays = P.random_rays(flat, N, seed=seed)
        # second-generation rays: start exactly on hit points of the first batch
        a = ctx.intersect_batch(rays)
        r2 = rays.copy()
        hit = a["hit"] != 0
        r2["o"][hit] = (rays["o"][hit] + rays["d"][hit] * a["distance"][hit, None]).astype(np.float32)
        rng = np.random.default_rng(seed + 1)
        d = rng.normal(size=(N, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        r2["d"] = d.astype(np.float32)
        for batch in (rays, r2):
            g = ctx.intersect_batch(batch)
            r = ry.intersect_batch(rb.bvh, rb.scene, batch)
            ok = P.hits_equal(g, r)
            if not ok:
                nb = int((g["hit"] != r["hit"]).sum() + ((g["hit"] != 0) & (r["hit"] != 0) & ((g["element"] != r["element"]) | (g["instance"] != r["instance"]) | (g["distance"] != r["distance"]))).sum())
                bad_total += nb
                print(name, seed, "MISMATCHES", nb)
    ctx.close()
    print(name, "ok", flush=True)
print("total mismatches", bad_total)
