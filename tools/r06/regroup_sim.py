#!/usr/bin/env python
"""Round 6: what would evicting the last walkers of a wavefront buy?  Logs the per-ray walk lengths of a few generations of a
streamed batch (ythip_get_stream_walk_steps) and replays ks_extend's wavefronts on the host: a wavefront costs its LONGEST lane's
steps; under the scheme, once fewer than A lanes are still walking (and at least B steps have been taken) the wavefront stops,
its walkers go to a continuation queue with their remaining steps, and the continuation passes regroup them 64 at a time
(the last pass runs to the end).  Prints wave-steps now / under the scheme, per generation.
  SCENE=configs4 GENS=2,40,120 python tools/r06/regroup_sim.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd")); sys.path.insert(0, ROOT)
import ythip as yt, bench


def waves(steps):
    n = (len(steps) + 63) // 64 * 64
    s = np.zeros(n, np.int64); s[:len(steps)] = steps
    return s.reshape(-1, 64)


def scheme(steps, A, B, passes):
    """-> (wave-steps of every pass, rays continued per pass)"""
    cost, moved = [], []
    cur = steps
    for p in range(passes):
        w = waves(cur)
        if p == passes - 1:
            cost.append(int(w.max(1).sum())); moved.append(0)
            break
        srt = -np.sort(-w, axis=1)           # descending per wave
        stop = np.maximum(srt[:, A - 1] if A <= 64 else srt[:, -1], B)   # the step at which fewer than A lanes are left (the A-th longest lane ends), not before B
        stop = np.minimum(stop, srt[:, 0])   # (never beyond the longest lane)
        cost.append(int(stop.sum()))
        rest = (w - stop[:, None]).ravel()
        rest = rest[rest > 0]
        moved.append(len(rest))
        if len(rest) == 0:
            break
        cur = rest
    return cost, moved


name = os.environ.get("SCENE", "configs4")
w = bench._workloads()[name]
flat = w["make"]()
spp = int(os.environ.get("SPP", w["spp"]))
ctx = bench.open_context(0, flat)
fm = int(os.environ.get("FASTMATH", "0"))
if fm == 2:
    ctx.make_own_bvh(flat)
p = yt.trace_params(sampler="path", resolution=w["resolution"], samples=1 << 30, batch=spp, fastmath=fm)
ctx.set_scheduler(1)
o, c = (int(x) for x in os.environ.get("VARIANT", "2:4").split(":"))
ctx.set_stream_options(order=o, cell_bits=c)
ctx.set_stream_groups(1)
ctx.set_stream_finish(0)
ctx.make_trace_state(flat, p)
ctx.trace_samples(p)
for gen in (int(x) for x in os.environ.get("GENS", "2,40,120").split(",")):
    ctx.set_profiling(1)
    ctx.stream_walk_steps(gen)
    ctx.trace_samples(p)
    g = ctx.stream_generations()
    steps = ctx.stream_walk_steps(gen, fetch=True)[:g[gen]].astype(np.int64)
    ctx.set_profiling(0)
    now = int(waves(steps).max(1).sum())
    print(f"{name} generation {gen}: {len(steps)} rays, lane steps {steps.sum()}, wave steps now {now} (evenness {steps.sum() / 64 / now:.3f}), "
          f"longest walk {steps.max()}, mean {steps.mean():.1f}, p50 {np.percentile(steps, 50):.0f} p90 {np.percentile(steps, 90):.0f} p99 {np.percentile(steps, 99):.0f}")
    for A, B, passes in ((16, 8, 2), (16, 8, 3), (24, 8, 3), (32, 8, 3), (32, 8, 4), (40, 16, 4), (48, 16, 5)):
        cost, moved = scheme(steps, A, B, passes)
        tot = sum(cost)
        print(f"   evict below {A:2d} lanes (after {B:2d} steps), {passes} passes: wave steps {tot} = x{now / tot:.3f}   per pass {cost}  continued {moved}")
ctx.close()
