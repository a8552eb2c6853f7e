#!/usr/bin/env python
"""Round 5 probe: does a frame rendered as G independent tile-column groups, each a chain of launches on its OWN stream, beat
one launch per batch?  (A launch must drain before the next one of its stream starts; with several chains the tail of one
chain's launch is filled by another chain's blocks.)   python tools/r05/overlap_probe.py [scene=plane] [res=1280] [spp=64] [steps=20]
Every group is a context of its own here (scene replicated): a probe, not the product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ythip as yt, scenes as ysc
from sharding import shard_frame
import parity as P

scene = sys.argv[1] if len(sys.argv) > 1 else "plane"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
fast = int(os.environ.get("FASTMATH", "0"))
flat = {"plane": ysc.plane_scene, "cornell1m": P.scene_cornell_1m, "cfg4": ysc.instanced_scene, "cfg5": ysc.hair_scene}[scene]()
p = yt.trace_params(sampler="path", resolution=res, bounces=8, clamp=10.0, samples=1 << 30, batch=spp, fastmath=fast)
w, h = yt.state_size(flat.cameras[0], p.resolution)
rngs = yt.make_rngs(p.seed, w * h)
dev = torch.device("cuda", 0)
digests = {}
for groups in (1, 2, 3, 4, 1, 2):
    ctxs, keep = [], []
    for g in range(groups):
        ctx = yt.Context(0)
        ctx.upload_scene(flat); ctx.make_trace_bvh(flat); ctx.make_trace_lights(flat)
        if fast == 2:
            ctx.make_own_bvh(flat)
        s = torch.cuda.Stream(device=dev)
        ctx.set_stream(s.cuda_stream)
        sh = shard_frame(w, h, groups, g, "columns")
        ctx.make_trace_state(flat, p, rows=sh.rows, cols=sh.cols, rngs=rngs)
        ctxs.append(ctx); keep.append((s, sh))
    torch.cuda.synchronize()
    for _ in range(5):
        for c in ctxs:
            c.trace_samples_async(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for c in ctxs:
            c.trace_samples_async(p)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # the frames agree whatever the grouping (pixels are independent): a digest of the gathered image
    img = np.zeros((h * w, 4), np.float32)
    for c, (s, sh) in zip(ctxs, keep):
        st = c.download_state()
        img[sh.pixels] = st["image"]
    digests[groups] = hash(img.tobytes())
    print(f"{scene} {w}x{h}x{spp} fastmath {fast}: {groups} group(s)  {dt * 1e3:8.3f} ms / step  {w * h * spp / dt / 1e6:10.1f} Msamples/s  image {digests[groups] & 0xffffffff:08x}", flush=True)
    for c in ctxs:
        c.close()
assert len(set(digests.values())) == 1, "the groupings disagree"
