#!/usr/bin/env python
"""Which of two development builds (tools/devbuild.sh) agrees with the compiled reference, and
where they differ:  python tools/diff_builds.py SCENE SAMPLER RES SPP NAME_A NAME_B
(TEST INFRASTRUCTURE: uses oracle/_ref as the checker.)"""
import os, subprocess, sys, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "--render":
    import parity as P
    scene, sampler, res, spp, out = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    flat = P.SCENES[scene]()
    p = P.yt.trace_params(sampler=sampler, resolution=res, samples=spp, batch=spp)
    if os.environ.get("USE_REF"):
        st = P.RefBundle(flat).render(p)
    else:
        ctx = P.gpu_context(flat)
        st = P.gpu_render(ctx, flat, p)
        ctx.close()
    pickle.dump(st, open(out, "wb"))
    sys.exit(0)
import numpy as np
scene, sampler, res, spp = sys.argv[1:5]
names = sys.argv[5:]
states = {}
for n in names + ["REF"]:
    env = dict(os.environ)
    if n == "REF":
        env["USE_REF"] = "1"
    else:
        env["YTHIP_LIB"] = os.path.join(ROOT, "build", "dev", f"libythip_{n}.so")
    out = f"/tmp/diff_{n}.pkl"
    subprocess.run([sys.executable, __file__, "--render", scene, sampler, res, spp, out], env=env, check=True)
    states[n] = pickle.load(open(out, "rb"))
ref = states["REF"]
w = ref["width"]
for n in names:
    st = states[n]
    print(f"== {n} vs reference")
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        a, b = np.asarray(st[k]), np.asarray(ref[k])
        d = np.any((a != b).reshape(len(b), -1), axis=1) if a.dtype.kind != "f" else np.any((a.view("u4") != b.view("u4")).reshape(len(b), -1), axis=1)
        idx = np.flatnonzero(d)
        print(f"   {k:7s} {len(idx):6d} pixels differ", [(int(i % w), int(i // w)) for i in idx[:6]])
        if len(idx) and k == "image":
            for i in idx[:3]:
                print("      pixel", (int(i % w), int(i // w)), "gpu", a[i], "ref", b[i], "rng equal", bool(np.all(st["rngs"][i] == ref["rngs"][i])))
