#!/usr/bin/env python
"""How long does an in-batch cancellation take?  The instanced scene at 1920x1080, a batch of 2048 spp
(seconds of work), the caller's stop flag raised after 0.3 s; prints the time from the flag to the return of
ythip_trace_samples.  Environment: YTHIP_STOP_MEM (0 ordinary device memory, 1 uncached, 2 fine-grained,
3 pinned host memory), YTHIP_LPT_PROBE."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd"))
import ythip as yt, scenes as ysc
flat = ysc.instanced_scene()
ctx = yt.Context(0)
ctx.upload_scene(flat); ctx.make_trace_bvh(flat); ctx.make_trace_lights(flat)
out = []
for rep in range(int(os.environ.get("REPS", "3"))):
    params = yt.trace_params(sampler="path", resolution=1920, samples=1 << 20, batch=2048)
    ctx.make_trace_state(flat, params)
    stop = np.zeros(1, np.int32)
    t_raise = [0.0]
    def raiser():
        time.sleep(0.3)
        t_raise[0] = time.perf_counter()
        stop[0] = 1
    th = threading.Thread(target=raiser); th.start()
    try:
        ctx.trace_samples(params, stop=stop)
        out.append(float("nan"))
    except yt.YthipError:
        out.append(1e3 * (time.perf_counter() - t_raise[0]))
    th.join()
print("STOP_MEM", os.environ.get("YTHIP_STOP_MEM", "default"), "LPT_PROBE", os.environ.get("YTHIP_LPT_PROBE", "default"),
      "cancel latency ms:", " ".join(f"{x:.1f}" for x in out))
