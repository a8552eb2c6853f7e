#!/usr/bin/env python
"""GPU vs the live reference (oracle/_ref), whole trace_state: how many pixels keep the
reference's rng stream, and is the state bit-identical?  Diagnostic for the libm port
(the parity tests proper are in tests/).   SCENES=... SAMPLERS=... RES= SPP= python tools/parity_report.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity as P
import ythip as yt
names = (os.environ.get("SCENES") or "cornellbox,plane,materials,instances,lines_points").split(",")
samplers = (os.environ.get("SAMPLERS") or "path").split(",")
res, spp = int(os.environ.get("RES", "128")), int(os.environ.get("SPP", "16"))
for name in names:
    if name.startswith("ref:"):
        flat = P.load_ref_scene(name[4:])
    elif name == "cornell1m":
        flat = P.scene_cornell_1m()
    else:
        flat = P.SCENES[name]()
    rb = P.RefBundle(flat)
    ctx = P.gpu_context(flat)
    for sampler in samplers:
        p = yt.trace_params(sampler=sampler, resolution=res, samples=spp, batch=spp)
        g = P.gpu_render(ctx, flat, p)
        r = rb.render(p)
        same_rng = float((g["rngs"] == r["rngs"]).all(axis=1).mean())
        ident = {k: g[k].tobytes() == r[k].tobytes() for k in ["image", "albedo", "normal", "hits", "rngs"]}
        px = float((g["image"] == r["image"]).all(axis=1).mean())
        print(f"{name:14s} {sampler:10s} {g['width']}x{g['height']}x{spp}: identical rng streams {same_rng:.6f}  "
              f"identical image pixels {px:.6f}  whole state identical: {all(ident.values())}  {ident if not all(ident.values()) else ''}", flush=True)
    ctx.close()
