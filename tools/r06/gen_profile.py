#!/usr/bin/env python
"""Round 6: the shape of a streamed batch — queue length per generation (ythip_get_stream_generations) next to the
generation's kernel times from a rocprofv3 kernel trace of the same process, when run under
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/r06/gen_profile.py
(the trace is joined afterwards by tools/r06/gen_join.py).  Prints the queue lengths, decimated."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd")); sys.path.insert(0, ROOT)
import ythip as yt, bench
name = os.environ.get("SCENE", "cfg2b")
w = bench._workloads()[name]
flat = w["make"]()
spp = int(os.environ.get("SPP", w["spp"]))
ctx = bench.open_context(0, flat)
p = yt.trace_params(sampler="path", resolution=w["resolution"], samples=1 << 30, batch=spp)
ctx.set_scheduler(1)
o, c = (int(x) for x in os.environ.get("VARIANT", "1:3").split(":"))
ctx.set_stream_options(order=o, cell_bits=c)
if os.environ.get("GROUPS"): ctx.set_stream_groups(int(os.environ["GROUPS"]))
if os.environ.get("FINISH"): ctx.set_stream_finish(int(os.environ["FINISH"]))
ctx.make_trace_state(flat, p)
ctx.trace_samples(p)
ctx.set_profiling(1); ctx.reset_stats()
ctx.trace_samples(p)
s = ctx.get_stats(); g = ctx.stream_generations(); info = ctx.stream_info()
print(f"{name} {spp} spp order {o} cells {c}: {s['trace_ms']:.3f} ms, {len(g)} generations, {g.sum() / 1e6:.1f} M rays, slots {ctx.npixels}")
np.save(os.environ.get("GEN_OUT", "/tmp/gen_rays.npy"), g)
print("generation: rays  ", "  ".join(f"{k}:{g[k]}" for k in range(0, len(g), max(1, len(g) // 24))))
for frac in (0.9, 0.5, 0.25, 0.1):
    print(f"generations with more than {frac:.2f} x slots queued: {(g > frac * ctx.npixels).sum()}  (rays in them: {g[g > frac * ctx.npixels].sum() / g.sum():.3f})")
