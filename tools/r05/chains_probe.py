#!/usr/bin/env python
"""Round 5 probe (needs tools/experiments/r05_batch_chains.patch applied): ythip_set_batch_chains(1 / 2) on one context — a loop of asynchronous batches, ms per step.
   python tools/r05/chains_probe.py [scene=plane] [res=1280] [spp=64] [steps=20]     (YTHIP_CHAIN_PRIORITY: the side stream's priority)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ythip as yt, scenes as ysc
import parity as P

scene = sys.argv[1] if len(sys.argv) > 1 else "plane"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
fast = int(os.environ.get("FASTMATH", "0"))
use_torch = os.environ.get("TORCH_STREAM", "0") == "1"
flat = {"plane": ysc.plane_scene, "cornell1m": P.scene_cornell_1m, "cfg4": ysc.instanced_scene, "cfg5": ysc.hair_scene}[scene]()
p = yt.trace_params(sampler="path", resolution=res, bounces=8, clamp=10.0, samples=1 << 30, batch=spp, fastmath=fast)
ctx = yt.Context(0)
ctx.upload_scene(flat); ctx.make_trace_bvh(flat); ctx.make_trace_lights(flat)
if fast == 2:
    ctx.make_own_bvh(flat)
if use_torch:
    import torch
    s = torch.cuda.Stream(device=torch.device("cuda", 0))
    ctx.set_stream(s.cuda_stream)
w, h = ctx.make_trace_state(flat, p)
for chains in (1, 2, 1, 2):
    ctx.set_batch_chains(chains)
    for _ in range(5):
        ctx.trace_samples_async(p)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.trace_samples_async(p)
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    print(f"{scene} {w}x{h}x{spp} fastmath {fast} torch stream {int(use_torch)} priority {os.environ.get('YTHIP_CHAIN_PRIORITY', 'greatest')}: "
          f"{chains} chain(s)  {dt * 1e3:8.3f} ms / step  {w * h * spp / dt / 1e6:10.1f} Msamples/s", flush=True)
ctx.close()
