#!/usr/bin/env python
"""Where does a step go?  Times trace_samples on BASELINE configs[1] for several
samplers (falsecolor = primary ray + minimal shading; eyelight = + material
evaluation; path = the full loop) and the bare traversal of the frame's primary
rays (k_intersect_batch).  Diagnostic only."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd"))
import ythip as yt, scenes as ysc

SCENE = os.environ.get("SCENE", "plane")
if SCENE == "plane":
    flat = ysc.plane_scene()
elif SCENE == "cfg4":  # BASELINE configs[3]: 10,000 instances of a 1,024-triangle sphere
    flat = ysc.instanced_scene()
elif SCENE == "cfg5":  # BASELINE configs[4]: 800,000 hair segments (the reference's make_hair scene: scenes.hair_scene)
    flat = ysc.hair_scene()
elif SCENE == "cornell1m":  # cfg2b: the Cornell box with 1M-triangle walls
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import parity as P
    flat = P.scene_cornell_1m()
elif SCENE == "cornell9m":  # bench.py's cache-exceeding scene: 8,987,066 wall triangles
    flat = ysc.cornell_1m_scene(ysc.load_scene(os.path.join(ROOT, "tests", "golden", "cornellbox.npz")), n=948)
elif SCENE.startswith("corpus:"):  # a scene of the reference's own test corpus (tests/golden/scenes): corpus:materials1 ...
    flat = ysc.load_corpus_scene(SCENE[7:])
else:
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import parity as P
    flat = P.SCENES[SCENE]()
ctx = yt.Context(0)
ctx.upload_scene(flat); ctx.make_trace_bvh(flat, bool(int(os.environ.get('HQ', '0')))); ctx.make_trace_lights(flat)  # HQ=1: highqualitybvh (split_sah)
if os.environ.get('TRAVERSAL'):
    ctx.set_traversal(os.environ['TRAVERSAL'])
if os.environ.get('SPECIALIZE'):  # 0: the general kernel class whatever the scene is (ythip_set_specialization)
    ctx.set_specialization(int(os.environ['SPECIALIZE']))
spp = int(os.environ.get('SPP', '64'))
RES = int(os.environ.get('RES', '1280'))
FAST = int(os.environ.get("FASTMATH", "0"))  # 1: the tolerance mode (params.fastmath), 2: the own-tree mode
if FAST == 2:
    ctx.make_own_bvh(flat)
    oi = ctx.own_bvh_info()
    print(f"[own tree] {oi['num_nodes']} nodes, {oi['bytes'] / 1e6:.1f} MB, build {oi['build_ms']:.1f} ms + bake {oi['bake_ms']:.1f} ms, depth {oi['max_depth']}")
for sampler in (os.environ.get("SAMPLERS") or "falsecolor,eyelight,naive,pathtest,path").split(","):
    p = yt.trace_params(sampler=sampler, resolution=RES, samples=1 << 30, batch=spp, fastmath=FAST)
    ctx.make_trace_state(flat, p)
    ctx.trace_samples(p)
    ctx.set_profiling(1); ctx.reset_stats()
    for _ in range(int(os.environ.get("LAUNCHES", "3"))):
        ctx.trace_samples(p)
    s = ctx.get_stats(); ctx.set_profiling(0)
    ms = s["trace_ms"] / s["trace_launches"]
    digest = ""
    if os.environ.get("DIGEST"):  # A/B builds must agree bit for bit: hash of the whole trace_state
        import hashlib
        d = ctx.download_state()
        h = hashlib.sha1()
        for k in sorted(d):
            if hasattr(d[k], "tobytes"): h.update(d[k].tobytes())
        digest = " state " + h.hexdigest()[:12]
    if os.environ.get("STACKSTATS"):  # -DYT_STACK_STATS builds: pushes / pushes beyond the LDS levels
        print(f"[stack] pushes {s['quads']}  beyond LDS {s['lines']}  ({100.0 * s['lines'] / max(1, s['quads']):.1f} %)")
    print(f"{SCENE} {sampler:10s} {ms:8.3f} ms/step  {ctx.npixels*spp/ms/1e3:8.1f} Msamples/s{digest}", flush=True)
