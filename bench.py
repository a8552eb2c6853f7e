#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on BASELINE.json's config.

  metric   Msamples/s (whole job) on the procedural 1M-triangle plane + constant
           environment, 1280x720, 64 spp, sampler=path, 8 bounces, clamp 10,
           default seed  (configs[1]; SURVEY.md §8d recipe cfg2)
  step     one trace_samples call rendering `--spp` (64) more samples for every
           pixel = 1280*720*64 = 58,982,400 camera paths
  N > 1    the frame's 16-pixel tile columns dealt round-robin across ranks
           (sharding.py — balanced, unlike contiguous row blocks whose top ranks
           would only see sky); every rank holds a full replica of scene+BVH and
           its slice of trace_state; no data-path collective; one RCCL all-gather
           of the framebuffer + un-permute per step (inside the timed region, on
           the kernel's stream: overlapping it with the next step's kernel on a
           side stream was measured slower — the persistent kernel holds every CU —
           see DESIGN.md §7).
           Primary line, "scaling": "weak" — the frame grows with N at the same
           camera (N x the pixels of configs[1]: 1280x720 per GPU, see
           weak_resolution()), so per-GPU work is fixed.
           The same line carries "configs2_strong": BASELINE configs[2] exactly
           (the 1280x720 frame split N ways, total work fixed), timed right after
           the primary region with the same K steps and fences.

Launch:  python bench.py [--gpus N --steps K --warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
                --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def measured_traffic():
    """HBM bytes per k_trace launch from the PMC passes of the same command
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled
    as MI355X_MICROARCH.md §HBM prescribes for gfx950), as committed by
    tools/prof.sh in profiles/traffic.json; None when no profile is present.
    Counters cannot be read from inside the timed run, so this is not live."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f)["k_trace"]["hbm_bytes_per_launch"]
    except Exception:
        return None


def measured_valu_busy():
    """Fraction of the VALU issue slots k_trace keeps busy, from the PMC pass of the
    same command committed under profiles/ (SQ_ACTIVE_INST_VALU x 4 cycles per wave64
    instruction / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs)); None when no profile is
    present.  What actually bounds the kernel: the working set is
    cache-resident, so the algorithmic byte rate exceeds the HBM peak."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc.json")) as f:
            d = json.load(f)
        k = [v for n, v in d.items() if "k_trace<0, 0, false, true, true>" in n][0]
        cycles = k["GRBM_GUI_ACTIVE"]["per_dispatch"] / 8.0  # summed over the 8 XCDs
        return round(k["SQ_ACTIVE_INST_VALU"]["per_dispatch"] * 4.0 / (cycles * 1024.0), 4)
    except Exception:
        return None


def cpu_baseline(flat, params_kw, budget_s=15.0):
    """The reference itself (oracle/_ref, g++ -O3, all host cores) timed on a
    bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import refyocto as ry
    import ythip as yt
    if not ry.available():
        return None
    scene = ry.RefScene.from_flat(flat)
    bvh, lights = ry.RefBvh(scene), ry.RefLights(scene)
    cores = ry.hardware_concurrency()
    # warm-up + rate probe with 1 spp, then size the sample to ~budget_s
    p = yt.trace_params(samples=1 << 20, batch=1, **params_kw)
    st = ry.RefState(scene, p)
    ry.trace_samples(st, scene, bvh, lights, p)
    t1 = ry.trace_samples(st, scene, bvh, lights, p)
    spp = int(max(1, min(64, budget_s / max(t1, 1e-3))))
    p = yt.trace_params(samples=1 << 20, batch=spp, **params_kw)
    t = ry.trace_samples(st, scene, bvh, lights, p)
    n = st.width * st.height * spp
    return {"value": round(n / t / 1e6, 3), "unit": "Msamples/s", "cores": cores,
            "kind": "reference",
            "sample": f"{st.width}x{st.height}x{spp}spp of the same scene/params "
                      f"({n} samples, {t:.2f} s, after a 2-pass warm-up), "
                      f"reference trace_samples via oracle/_ref (g++ -O3, std::async x{cores})"}


def other_configs(device, args):
    """The other single-GPU BASELINE configs through the same path, short runs (their
    parity lives in tests/test_gpu_baseline_configs.py): cfg2b — the north star's
    "1M-triangle Cornell-box-style scene" (every ray hits, area-light
    pdf walks) at 1024x1024x64spp — and configs[3], 10,000 instances of a
    1,024-triangle mesh at 1920x1080x256spp.  One counting launch + 1 warm-up + 2 timed
    steps each; algorithmic bytes by the same formula as the headline."""
    import ythip as yt
    import scenes as ysc
    res = []
    cornell = os.path.join(ROOT, "tests", "golden", "cornellbox.npz")  # make_cornellbox() as exported
    for name, make, resolution, spp in [
            ("cfg2b: Cornell box with 998,586 wall triangles, 1 area light",
             lambda: ysc.cornell_1m_scene(ysc.load_scene(cornell)), 1024, 64),
            ("configs[3]: 10,000 instances x 1,024-triangle sphere + constant env",
             ysc.instanced_scene, 1920, 256)]:
        flat = make()
        ctx = yt.Context(device)
        ctx.upload_scene(flat)
        ctx.make_trace_bvh(flat)
        ctx.make_trace_lights(flat)
        p = yt.trace_params(sampler="path", resolution=resolution, bounces=8, clamp=10.0,
                            samples=1 << 30, batch=spp)
        w, h = ctx.make_trace_state(flat, p)
        ctx.set_profiling(2)
        ctx.reset_stats()
        ctx.trace_samples(p)
        cnt = ctx.get_stats()
        ctx.set_profiling(0)
        ctx.trace_samples(p)
        ctx.set_profiling(1)
        ctx.reset_stats()
        steps = 2
        for _ in range(steps):
            ctx.trace_samples(p)
        st = ctx.get_stats()
        ctx.set_profiling(0)
        ctx.close()
        ms = st["trace_ms"] / steps
        nsamp = max(cnt["samples"], 1)
        bps = yt.algorithmic_bytes(cnt) / nsamp
        res.append({"workload": f"{name}, {w}x{h}x{spp}spp, sampler=path bounces=8 clamp=10",
                    "value": round(w * h * spp / ms / 1e3, 3), "unit": "Msamples/s",
                    "ms_per_step": round(ms, 3), "steps": steps,
                    "bytes_per_sample": round(bps, 1),
                    "achieved_GBps": round(bps * w * h * spp / (ms * 1e-3) / 1e9, 1),
                    "per_sample": {"rays": round(cnt["rays"] / nsamp, 3),
                                   "nodes": round(cnt["nodes"] / nsamp, 3),
                                   "triangles": round(cnt["triangles"] / nsamp, 3),
                                   "instances": round(cnt["instances"] / nsamp, 3),
                                   "shades": round(cnt["shades"] / nsamp, 3)}})
    return res


def weak_resolution(base, world, tile=16):
    """Width of the weak-scaling frame: the same camera at `world` x the pixels of
    the `base`-wide frame, i.e. base * sqrt(world), rounded to a multiple of
    tile * world so every rank owns the same number of 16-pixel tile columns —
    preferring a multiple of 8 (then 4, 2) tile columns per rank (measured: 57- and
    58-column slices run 3-6 % slower per pixel than 56- and 80-column ones) — as
    long as the pixel count stays within 5 % of world x base.
    1280 -> 1792 / 2560 / 3584 for 2 / 4 / 8 ranks: 1.96x / 4x / 7.84x the pixels."""
    for m in (8, 4, 2, 1):
        q = tile * world * m
        w = max(q, int(round(base * world ** 0.5 / q)) * q)
        if abs(w * w / (base * base * world) - 1) <= 0.05 or m == 1:
            return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--resolution", type=int, default=1280)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N=1: skip the short runs of the other BASELINE configs (other_configs)")
    ap.add_argument("--sharding", choices=["columns", "rows"], default="columns")
    ap.add_argument("--scaling", choices=["weak", "strong", "both"], default="both",
                    help="N > 1: weak = frame grown to N x the pixels of configs[1] (primary "
                         "line); strong = configs[2], the 1280x720 frame split N ways; both = "
                         "weak as the primary line + configs2_strong inside it")
    ap.add_argument("--overlap-gather", action="store_true",
                    help="experiment: gather a snapshot of the frame on a side stream while the "
                         "next step's kernel runs (measured slower than the default)")
    ap.add_argument("--traversal", choices=["auto", "binary", "wide"], default="auto")
    ap.add_argument("--rehearse-gather", action="store_true",
                    help="N=1 only: create a world_size-1 RCCL group and run the framebuffer "
                         "gather path anyway (checks the stream handling on a 1-GPU box)")
    ap.add_argument("--as-rank", default=None, metavar="R/N",
                    help="single-GPU experiment: render only the slice rank R of N would "
                         "(no gather); the JSON line then describes that slice")
    args = ap.parse_args()

    import torch
    import ythip as yt
    import scenes as ysc
    from sharding import FrameGather, shard_frame

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("YTHIP_DIST_BACKEND", "nccl")  # "gloo": rehearse the N > 1 path on ONE GPU
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback for the measured path")
    if backend != "nccl":  # rehearsal: several ranks may share a device
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    gathering = world > 1 or args.rehearse_gather
    if gathering:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29633")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    # ---- workload: BASELINE.json configs[1] --------------------------------
    flat = ysc.plane_scene()  # 1,000,000 triangles, 501,501 vertices
    params_kw = dict(sampler="path", resolution=args.resolution, bounces=8, clamp=10.0)
    params = yt.trace_params(samples=1 << 30, batch=args.spp, **params_kw)
    t0 = time.time()
    ctx = yt.Context(local)
    ctx.upload_scene(flat)
    ctx.make_trace_bvh(flat)
    ctx.make_trace_lights(flat)
    ctx.set_traversal(args.traversal)
    setup_s = time.time() - t0
    build_info = ctx.bvh_build_info()
    dev = torch.device("cuda", local)
    stream = torch.cuda.Stream(device=dev)  # non-null: the kernels run here
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    comm = torch.cuda.Stream(device=dev) if gathering and args.overlap_gather else None

    def fence():
        if gathering:
            dist.barrier()
        torch.cuda.synchronize()  # device-wide: kernel stream, gather stream, RCCL's own

    def run(resolution, want_roofline):
        """Render `--steps` timed steps of the frame of the given resolution, sharded
        over the ranks; returns (seconds [max over ranks], frame size, pixels of this
        rank, counting-pass stats, timing stats)."""
        p = yt.trace_params(samples=1 << 30, batch=args.spp, **dict(params_kw, resolution=resolution))
        w, h = yt.state_size(flat.cameras[0], p.resolution)
        if args.as_rank:
            vr, vn = (int(x) for x in args.as_rank.split("/"))
            shard = shard_frame(w, h, vn, vr, args.sharding)
        else:
            shard = shard_frame(w, h, world, rank, args.sharding)
        rngs = yt.make_rngs(p.seed, w * h)
        ctx.make_trace_state(flat, p, rows=shard.rows, cols=shard.cols, rngs=rngs)
        npix = shard.npixels
        assert npix == ctx.npixels
        # state arrays live in torch tensors so the RCCL gather runs on them directly
        image = torch.zeros(npix, 4, device=dev)
        albedo = torch.zeros(npix, 3, device=dev)
        normal = torch.zeros(npix, 3, device=dev)
        hits = torch.zeros(npix, dtype=torch.int32, device=dev)
        trng = torch.from_numpy(shard.take(rngs).view(np.int64).copy()).to(dev)
        torch.cuda.synchronize()
        ctx.bind_device_state(image.data_ptr(), albedo.data_ptr(), normal.data_ptr(),
                              hits.data_ptr(), trng.data_ptr())
        gather = FrameGather(dist, w, h, 4, dev, mode=args.sharding,
                             always=args.rehearse_gather) if gathering else None
        snapshot = torch.empty_like(image) if comm is not None else None

        def step():
            ctx.trace_samples_async(p)
            if not gathering:
                return
            # framebuffer gather over RCCL/xGMI (§8e), once per batch
            if comm is None:
                gather.frame(image)
                return
            # the next step's kernel updates `image` in place: gather a snapshot (a
            # 16 B/pixel device copy) on the side stream while that kernel runs
            stream.wait_stream(comm)   # the previous gather has read the snapshot
            snapshot.copy_(image)
            comm.wait_stream(stream)
            with torch.cuda.stream(comm):
                gather.frame(snapshot)

        # algorithmic work of one step (counting pass, untimed)
        stats_count = None
        if want_roofline:
            ctx.set_profiling(2)
            ctx.reset_stats()
            step()
            fence()
            stats_count = ctx.get_stats()
            ctx.set_profiling(0)
        for _ in range(args.warmup):
            step()
        fence()
        # timed region: exactly K steps
        ctx.set_profiling(1 if want_roofline else 0)  # hipEvents around the k_trace launches
        ctx.reset_stats()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        stats_time = ctx.get_stats()
        ctx.set_profiling(0)
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        if gathering:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # the state's buffers stay referenced until the context is re-bound
        keep.append((image, albedo, normal, hits, trng, snapshot, gather))
        return float(tmax.item()), (w, h), npix, stats_count, stats_time

    keep = []
    weak = world > 1 and args.scaling in ("weak", "both") and not args.as_rank
    resolution = weak_resolution(args.resolution, world) if weak else args.resolution
    dt, (w, h), npix, stats_count, stats_time = run(resolution, not args.no_roofline)

    total_samples = (npix if args.as_rank else w * h) * args.spp * args.steps
    value = total_samples / dt / 1e6
    if world == 1:
        what = "BASELINE configs[1]"
    elif weak:
        what = (f"configs[1] per GPU: the configs[1] camera at {world}x the pixels of "
                f"{args.resolution}x{args.resolution * 9 // 16}")
    else:
        what = "BASELINE configs[2]: the configs[1] frame split across the ranks"

    out = {
        "metric": "Msamples/s", "value": round(value, 3), "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak" if (weak or world == 1) else "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"1M-triangle plane + constant env, {w}x{h}x{args.spp}spp, "
                               f"sampler=path bounces=8 clamp=10 ({what})"
                               + (f" — ONLY the slice of rank {args.as_rank}" if args.as_rank else ""),
                   "triangles": int(flat.shapes[0]["num_triangles"]),
                   "resolution": [w, h], "spp": args.spp, "pixels_per_rank": npix,
                   "sharding": (f"{args.sharding}/{world}" if world > 1 else "none")
                               if not args.as_rank else f"{args.sharding} {args.as_rank}",
                   "framebuffer_gather": ("none" if not gathering else
                                          "rccl all_gather + un-permute per step, "
                                          + ("on the kernel stream" if comm is None else
                                             "on a side stream (overlaps the next step)")),
                   "setup_s": round(setup_s, 3),
                   # make_trace_bvh: the 1M-triangle tree is built ON THE DEVICE (identical to
                   # the reference's tree); wall ms of tree construction / baking the traversal layout
                   "bvh_build": {"builder": "device" if build_info["device_trees"] else "host",
                                 "device_kernels_ms": round(build_info["device_ms"], 3),
                                 "build_ms": round(build_info["build_ms"], 3),
                                 "bake_ms": round(build_info["bake_ms"], 3)}},
    }
    if rank == 0 and stats_count is not None and stats_time["trace_launches"] > 0:
        # The dominant (only) kernel is k_trace: one launch = one step.  Its
        # ALGORITHMIC bytes are SURVEY.md §8(d)'s per-unit figures x the units
        # counted in one step (all stages: traversal + shading + trace_state).
        launches_per_step = stats_time["trace_launches"] / args.steps
        bytes_step = yt.algorithmic_bytes(stats_count)
        k_ms = stats_time["trace_ms"] / stats_time["trace_launches"]
        bytes_per_launch = bytes_step / launches_per_step
        achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
        nsamp = max(stats_count["samples"], 1)
        out["roofline"] = {
            "bound": "hbm", "kernel": "k_trace", "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": measured_traffic() if world == 1 and not args.as_rank else None,
            "valu_busy": measured_valu_busy() if world == 1 and not args.as_rank else None,
            "launch_ms_avg": round(k_ms, 4), "launches_per_step": launches_per_step,
            "bytes_per_launch": int(bytes_per_launch),
            "traversal_bytes_per_launch": int(yt.traversal_bytes(stats_count) / launches_per_step),
            "per_sample": {"rays": round(stats_count["rays"] / nsamp, 3),
                           "nodes": round(stats_count["nodes"] / nsamp, 3),
                           "triangles": round(stats_count["triangles"] / nsamp, 3),
                           "instances": round(stats_count["instances"] / nsamp, 3),
                           "shades": round(stats_count["shades"] / nsamp, 3),
                           "bytes_all_stages": round(bytes_step / nsamp, 1)},
        }
        if world > 1:
            out["roofline"]["note"] = "rank 0's launches (its slice of the frame)"
    if weak and args.scaling == "both":
        # BASELINE configs[2] exactly: the 1280x720 frame split N ways (total work
        # fixed), same K steps between the same fences.  Reported, never required:
        # a failure here (identical on every rank) must not cost the primary line.
        try:
            dt2, (w2, h2), npix2, _, _ = run(args.resolution, False)
            out["configs2_strong"] = {
                "value": round(w2 * h2 * args.spp * args.steps / dt2 / 1e6, 3), "unit": "Msamples/s",
                "ms_per_step": round(dt2 / args.steps * 1e3, 3), "scaling": "strong",
                "resolution": [w2, h2], "spp": args.spp, "pixels_per_rank": npix2,
                "note": "a pixel's samples are sequential by contract (its PCG stream and running "
                        "mean), so one pixel's 64-sample chain (~1.3 ms) bounds the step however "
                        "few pixels a GPU holds (DESIGN.md §7)"}
        except Exception as e:
            out["configs2_strong"] = {"error": str(e)}
    if rank == 0 and world == 1 and not args.as_rank and not args.no_other_configs:
        ctx.close()
        try:
            out["other_configs"] = other_configs(local, args)
        except Exception as e:  # reported, never required
            out["other_configs"] = {"error": str(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(flat, params_kw)
        except Exception as e:  # the baseline is reported, never required
            out["cpu_baseline"] = {"error": str(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if gathering:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
