#!/usr/bin/env python
"""Round 6: the streaming scheduler (csrc/yt_stream.h) against the fused kernel on bench.py's workloads.
Per workload: the fused kernel's time + whole-state digest, then the streaming scheduler in the variants of VARIANTS
(order:cells[:phased[:groups[:finish]]]; order 0 octant major, 1 cell major, 2 unsorted) — time, digest (must equal), generations, and how even
the walks of a wavefront are (sum of lane steps / 64 x longest lane; the fused kernel's figure is in profiles/r03_traversal_experiments.txt).

  SCENES=cfg2b,configs4 SPP=64 VARIANTS=0:4,1:4,2:4 LAUNCHES=2 python tools/r06/stream_ab.py
FASTMATH=1 / 2: both sides in the tolerance / own-tree mode (the scheduler's kernels of yt_fast.hip / yt_owntree.hip).
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd"))
sys.path.insert(0, ROOT)
import ythip as yt  # noqa: E402
import bench  # noqa: E402


def digest(ctx):
    d = ctx.download_state()
    h = hashlib.sha1()
    for k in sorted(d):
        if hasattr(d[k], "tobytes"):
            h.update(d[k].tobytes())
    return h.hexdigest()[:12]


def timed(ctx, flat, p, launches, prof=True):
    ctx.make_trace_state(flat, p)
    ctx.trace_samples(p)  # warm-up (and the fused kernel's tile costs / pool decision)
    if launches > 1:
        ctx.trace_samples(p)
    ctx.set_profiling(1 if prof else 0)
    ctx.reset_stats()
    for _ in range(launches):
        ctx.trace_samples(p)
    s = ctx.get_stats()
    ctx.set_profiling(0)
    return s["trace_ms"] / launches


def main():
    scenes = (os.environ.get("SCENES") or "cfg2b,configs3,configs4,features1,materials1").split(",")
    variants = [tuple(float(x) for x in v.split(":")) for v in (os.environ.get("VARIANTS") or "0:4,1:4,2:4").split(",")]
    launches = int(os.environ.get("LAUNCHES", "2"))
    for name in scenes:
        w = bench._workloads()[name]
        flat = w["make"]()
        spp = int(os.environ.get("SPP", w["spp"]))
        res = int(os.environ.get("RES", w["resolution"]))
        ctx = bench.open_context(0, flat)
        fm = int(os.environ.get("FASTMATH", "0"))
        if fm == 2:
            ctx.make_own_bvh(flat)
        p = yt.trace_params(sampler=os.environ.get("SAMPLER", "path"), resolution=res, samples=1 << 30, batch=spp, fastmath=fm)
        ctx.set_scheduler(0)
        ms0 = timed(ctx, flat, p, launches, prof=os.environ.get("PROF", "1") != "0")
        # the digest of a fixed number of batches from a fresh state
        ctx.make_trace_state(flat, p)
        ctx.trace_samples(p)
        d0 = digest(ctx)
        npix = ctx.npixels
        print(f"{name:10s} {res}x{spp}spp fastmath {fm} fused            {ms0:9.3f} ms {npix * spp / ms0 / 1e3:9.1f} Msamples/s  state {d0}", flush=True)
        for v in variants:
            order, cells = int(v[0]), int(v[1])
            phased = int(v[2]) if len(v) > 2 else -1
            groups = int(v[3]) if len(v) > 3 else 2
            finish = int(v[4]) if len(v) > 4 else 250  # the tail kernel's threshold, thousandths of the path slots (0: off)
            ctx.set_stream_finish(finish)
            ctx.set_scheduler(1)
            ctx.set_stream_options(order=order, cell_bits=cells, phased=phased)
            ctx.set_stream_groups(groups)
            ms = timed(ctx, flat, p, launches, prof=os.environ.get("PROF", "1") != "0")
            info = ctx.stream_info()
            ctx.make_trace_state(flat, p)
            ctx.trace_samples(p)
            d = digest(ctx)
            even = info["lane_steps"] / max(1, info["wave_steps"])
            print(f"{name:10s} stream order {order} cells {cells} phased {phased:2d} groups {info['groups']} finish {finish:4d} ({info['finish_rays'] / 1e3:.0f} k) {ms:9.3f} ms {npix * spp / ms / 1e3:9.1f} Msamples/s  x{ms0 / ms:.3f}  "
                  f"state {d} {'OK' if d == d0 else 'DIFFERENT'}  generations {info['generations']} (+{info['launched'] - info['generations']} empty)  "
                  f"rays {info['rays'] / 1e6:.1f} M  walk evenness {even:.3f}", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
