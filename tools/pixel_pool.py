#!/usr/bin/env python
"""The pixel pool (ythip_set_pixel_pool) on the bench scenes: ms per step of `path` with the pool off, forced on, and
chosen by the library's own measurement (with what it measured and decided), plus the digest of the whole trace_state
(equal digests = bit-identical results).   SCENES="plane cornell1m cfg4 cfg5" python tools/pixel_pool.py"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = {"plane": (1280, 64), "cornell1m": (1024, 16), "cornell1m_hd": (1920, 16), "cfg4": (1920, 32), "cfg5": (1280, 16), "cornell9m": (1024, 16)}

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    scene, mode = sys.argv[2], int(sys.argv[3])
    os.environ["SCENE"] = scene.replace("_hd", "")
    os.environ["SAMPLERS"] = ""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.argv = sys.argv[:1]
    import importlib.util
    spec = importlib.util.spec_from_file_location("sampler_scene", os.path.join(ROOT, "tools", "sampler_times.py"))
    src = open(os.path.join(ROOT, "tools", "sampler_times.py")).read().split("spp = int(os.environ.get('SPP'")[0]
    g = {"__file__": os.path.join(ROOT, "tools", "sampler_times.py"), "__name__": "scene"}
    exec(compile(src, "sampler_times.py", "exec"), g)  # builds `flat`, `ctx` (scene, bvh, lights resident)
    yt, ctx, flat = g["yt"], g["ctx"], g["flat"]
    res, spp = SIZES[scene]
    ctx.set_pixel_pool(mode)
    p = yt.trace_params(sampler="path", resolution=res, samples=1 << 30, batch=spp)
    ctx.make_trace_state(flat, p)
    times = []
    for step in range(8):
        ctx.set_profiling(1), ctx.reset_stats()
        ctx.trace_samples(p)
        s = ctx.get_stats()
        times.append(s["trace_ms"])
    d = ctx.download_state()
    h = hashlib.sha1()
    for k in sorted(d):
        if hasattr(d[k], "tobytes"):
            h.update(d[k].tobytes())
    info = ctx.pixel_pool_info()
    tail = sorted(times[4:])[len(times[4:]) // 2]
    print(f"{scene:12s} pool={mode}  steps(ms) {' '.join(f'{t:7.2f}' for t in times)} | median of the last 4: {tail:7.2f} ms "
          f"{ctx.npixels * spp / tail / 1e3:8.1f} Msamples/s | decided={info['decided']} on={info['on']} "
          f"plain {info['plain_ms_per_sample']:.4f} pool {info['pool_ms_per_sample']:.4f} ms/sample | state {h.hexdigest()[:12]}", flush=True)
    sys.exit(0)

for scene in (os.environ.get("SCENES") or "plane cornell1m cfg4 cfg5").split():
    for mode in (0, 2, 1):
        subprocess.run([sys.executable, __file__, "--one", scene, str(mode)], check=False)
