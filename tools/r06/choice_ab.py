#!/usr/bin/env python
"""Round 6: ythip_set_scheduler(ctx, 2) — the measured choice between the fused kernel and the streaming scheduler — on bench.py's
workloads: what it decides, on what numbers, and the steady-state time per batch next to the fused kernel's.
  SCENES=cfg2b,configs1 python tools/r06/choice_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd")); sys.path.insert(0, ROOT)
import ythip as yt, bench
from stream_ab import digest

for name in (os.environ.get("SCENES") or "configs1,cfg2b,configs3,configs4,cornell9m,materials1,features1").split(","):
    w = bench._workloads()[name]
    flat = w["make"]()
    spp = int(os.environ.get("SPP", w["spp"]))
    fm = int(os.environ.get("FASTMATH", "0"))
    p = yt.trace_params(sampler=os.environ.get("SAMPLER", "path"), resolution=w["resolution"], samples=1 << 30, batch=spp, fastmath=fm)
    out = {}
    for sched in (0, 2):
        ctx = bench.open_context(0, flat)
        if fm == 2:
            ctx.make_own_bvh(flat)
        ctx.set_scheduler(sched)
        ctx.make_trace_state(flat, p)
        for _ in range(7):  # plain, pool, [fused timed, streamed, streamed timed, decided]
            ctx.trace_samples(p)
        info = ctx.stream_info()
        ctx.set_profiling(1); ctx.reset_stats()
        for _ in range(3):
            ctx.trace_samples(p)
        ms = ctx.get_stats()["trace_ms"] / 3
        ctx.set_profiling(0)
        out[sched] = (ms, digest(ctx), info)
        ctx.close()
    (ms0, d0, _), (ms2, d2, info) = out[0], out[2]
    print(f"{name:10s} {os.environ.get('SAMPLER', 'path')} fastmath {fm} fused {ms0:9.3f} ms | choice: state {info['choice_state']} streamed {info['choice_streamed']} "
          f"(probes: fused {info['fused_ms_per_sample'] * spp:9.3f} ms, streamed {info['stream_ms_per_sample'] * spp:9.3f} ms) -> {ms2:9.3f} ms  x{ms0 / ms2:.3f}  "
          f"state {d2} {'OK' if d0 == d2 else 'DIFFERENT'}", flush=True)
