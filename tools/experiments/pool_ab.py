#!/usr/bin/env python
"""A/B of the two schedulers on one GPU: for every scene, render the same batch with
k_trace (pool mode 0) and k_pool (mode 1), compare the whole trace_state bit for bit
and print the step times + k_pool's own statistics.  Diagnostic; the parity tests
proper live in tests/test_gpu_pool.py.

  SCENES=plane,cfg4,cfg5,cornell1m,cornellbox,...  RES= SPP= STEPS=  python tools/pool_ab.py
  POOL_CFGS="waves:target:refill:shademin:tilemul;..."   k_pool configurations to try
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ythip as yt
import scenes as ysc

DEFAULT_SIZES = {"plane": (1280, 64), "cfg4": (1920, 32), "cfg5": (1280, 16), "cornell1m": (1024, 16)}


def make_scene(name):
    if name == "plane":
        return ysc.plane_scene()
    if name == "cfg4":
        return ysc.instanced_scene()
    if name == "cfg5":
        import test_gpu_baseline_configs as T
        return T.hair_scene()
    import parity as P
    if name == "cornell1m":
        return P.scene_cornell_1m()
    return P.SCENES[name]()


def digest(state):
    h = hashlib.sha1()
    for k in sorted(state):
        if hasattr(state[k], "tobytes"):
            h.update(state[k].tobytes())
    return h.hexdigest()[:16]


def render(flat, sampler, res, spp, steps, pool, cfg=None, want_state=True):
    ctx = yt.Context(0)
    ctx.upload_scene(flat)
    ctx.make_trace_bvh(flat)
    ctx.make_trace_lights(flat)
    if pool:
        w, t, r, s, m = cfg
        ctx.set_pool(1, w, t, r, s, m)
    else:
        ctx.set_pool(0)
    p = yt.trace_params(sampler=sampler, resolution=res, samples=1 << 30, batch=spp)
    ctx.make_trace_state(flat, p)
    ctx.trace_samples(p)  # warm-up (also part of the compared state)
    ctx.set_profiling(1)
    ctx.reset_stats()
    for _ in range(steps):
        ctx.trace_samples(p)
    s = ctx.get_stats()
    ctx.set_profiling(0)
    ms = s["trace_ms"] / max(s["trace_launches"], 1)
    st = ctx.download_state() if want_state else None
    ps = ctx.pool_stats() if pool else None
    n = ctx.npixels
    ctx.close()
    return ms, (digest(st) if st is not None else ""), ps, n


def main():
    names = (os.environ.get("SCENES") or "cornellbox,plane,cornell1m,cfg4").split(",")
    sampler = os.environ.get("SAMPLER", "path")
    steps = int(os.environ.get("STEPS", "2"))
    cfgs = []
    for c in (os.environ.get("POOL_CFGS") or "0:0:8:64:1").split(";"):
        cfgs.append(tuple(int(x) for x in c.split(":")))
    for name in names:
        flat = make_scene(name)
        res, spp = DEFAULT_SIZES.get(name, (256, 8))
        res = int(os.environ.get("RES", res))
        spp = int(os.environ.get("SPP", spp))
        ms0, d0, _, n = render(flat, sampler, res, spp, steps, False)
        print(f"{name:12s} {sampler:8s} k_trace                      {ms0:9.3f} ms/step "
              f"{n * spp / ms0 / 1e3:9.1f} Msamples/s  state {d0}", flush=True)
        for cfg in cfgs:
            t0 = time.time()
            ms1, d1, (ssum, smax), _ = render(flat, sampler, res, spp, steps, True, cfg)
            ok = "IDENTICAL" if d1 == d0 else "DIFFERENT"
            line = (f"{name:12s} {sampler:8s} k_pool {':'.join(map(str, cfg)):20s} {ms1:9.3f} ms/step "
                    f"{n * spp / ms1 / 1e3:9.1f} Msamples/s  state {d1} {ok}  x{ms0 / ms1:.2f}")
            if ssum["watchdog"]:
                line += f"  WATCHDOG sum {ssum['watchdog']:#x} max {smax['watchdog']:#x}"
            if ssum["rounds"]:
                line += (f" | lanes/round {ssum['active'] / ssum['rounds']:.1f} paths/shade "
                         f"{ssum['shaded'] / max(ssum['shades'], 1):.1f} step-util "
                         f"{ssum['steps'] / max(64 * ssum['wsteps'], 1):.3f} iters max {smax['iters']} "
                         f"tiles max {smax['tiles']} Mcycles sum {ssum['cycles'] / 1e6:.0f} max {smax['cycles'] / 1e6:.2f}")
            print(line, flush=True)


if __name__ == "__main__":
    main()
