#!/usr/bin/env python
"""Round 6, first measurement: what would a between-bounce SORT buy the traversal?

The streaming scheduler the verdict asks for regroups rays between bounces by (direction octant, origin cell).
Before building it, this probe bounds its gain with the kernel that already exists: k_intersect_batch (the wide
walk and nothing else) on REAL bounce rays of a workload — the camera rays' hit points with a random direction
in the hemisphere that faces the viewer, then those rays' hit points again —, in four orders:

  tile     64 consecutive rays = a 16 x 4 pixel tile: what a wavefront of the fused kernel walks together
  random   a shuffle: the worst case
  oct+cell counting-sort key (direction octant, Morton code of the origin's cell of the scene box), cells^3 cells
  cell+oct the same with the cell as the major key

  SCENE=cfg2b|configs3|configs4|features1|materials1 [RES=..] python tools/r06/sorted_extend_probe.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd"))
sys.path.insert(0, ROOT)
import ythip as yt  # noqa: E402
import bench  # noqa: E402


def tile_order(w, h):
    """permutation: position in tile order -> pixel index (row-major), 16 x 4 tiles, edge tiles ragged"""
    j, i = np.mgrid[0:h, 0:w]
    tiles_x = (w + 15) // 16
    key = ((j // 4) * tiles_x + i // 16) * 64 + (j % 4) * 16 + (i % 16)
    return np.argsort(key.ravel(), kind="stable")


def part1by2(x):
    x = x.astype(np.uint64) & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x


def sort_keys(rays, lo, hi, cells, dirbits=0):
    o, d = rays["o"], rays["d"]
    c = np.clip(((o - lo) / np.maximum(hi - lo, 1e-20) * cells).astype(np.int64), 0, cells - 1)
    morton = part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2)
    octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << 1) | ((d[:, 2] < 0).astype(np.uint64) << 2))
    if dirbits:  # finer direction bins: the major axis and a dirbits x dirbits grid on the cube face
        a = np.abs(d)
        major = np.argmax(a, axis=1)
        idx = np.arange(len(d))
        u = d[idx, (major + 1) % 3] / a[idx, major]
        v = d[idx, (major + 2) % 3] / a[idx, major]
        ub = np.clip(((u * 0.5 + 0.5) * dirbits).astype(np.int64), 0, dirbits - 1)
        vb = np.clip(((v * 0.5 + 0.5) * dirbits).astype(np.int64), 0, dirbits - 1)
        face = major * 2 + (d[idx, major] < 0)
        octant = (face * dirbits * dirbits + ub * dirbits + vb).astype(np.uint64)
    return octant, morton


def bounce(rng, rays, hits):
    """next generation: from the hit points, uniform directions in the hemisphere facing the previous ray's origin"""
    m = hits["hit"] != 0
    o = rays["o"][m] + rays["d"][m] * hits["distance"][m, None]
    d = rng.normal(size=(int(m.sum()), 3)).astype("f4")
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    flip = np.sum(d * -rays["d"][m], axis=1) < 0
    d[flip] = -d[flip]
    out = np.zeros(len(o), yt.ray_dt)
    out["o"], out["d"], out["tmin"], out["tmax"] = o.astype("f4"), d, 1e-4, np.finfo("f4").max
    return out, np.nonzero(m)[0]


def timed(ctx, rays, reps=3):
    ctx.intersect_batch(rays[: min(len(rays), 65536)])
    ctx.set_profiling(1)
    ctx.reset_stats()
    for _ in range(reps):
        h = ctx.intersect_batch(rays)
    s = ctx.get_stats()
    ctx.set_profiling(0)
    return s["trace_ms"] / s["trace_launches"], h


def main():
    name = os.environ.get("SCENE", "cfg2b")
    w = bench._workloads()[name]
    flat = w["make"]()
    res = int(os.environ.get("RES", w["resolution"]))
    ctx = bench.open_context(0, flat)
    p = yt.trace_params(resolution=res, samples=1, batch=1)
    width, height = ctx.make_trace_state(flat, p)
    cam = ctx.camera_rays(p)
    perm = tile_order(width, height)
    cam = cam[perm]  # tile order from here on: a ray's index is its slot
    rng = np.random.default_rng(11)
    lo = hi = None  # the sort's cell grid spans the batch's own origins (on the device: the TLAS root box)
    print(f"# {name} {width}x{height}: {len(cam)} camera rays", flush=True)
    ms0, h0 = timed(ctx, cam)
    print(f"gen0 camera rays   tile   {ms0:8.3f} ms {len(cam) / ms0 / 1e3:9.1f} Mrays/s  hit {h0['hit'].mean():.3f}")
    gens = []
    rays, hits = cam, h0
    for g in (1, 2, 3):
        rays, src = bounce(rng, rays, hits)
        if len(rays) < 1000:
            break
        ms, hits = timed(ctx, rays)
        gens.append((g, rays, ms, hits))
    for g, rays, ms_tile, hits in gens:
        n = len(rays)
        lo, hi = rays["o"].min(0), rays["o"].max(0)
        ref = hashlib.sha1(np.sort(hits, order=["instance", "element", "distance"]).tobytes()).hexdigest()[:10]
        print(f"gen{g} bounce rays   tile   {ms_tile:8.3f} ms {n / ms_tile / 1e3:9.1f} Mrays/s  hit {hits['hit'].mean():.3f}  x1.00", flush=True)
        orders = {"random": rng.permutation(n)}
        for cells in (8, 16, 32, 64):
            octant, morton = sort_keys(rays, lo, hi, cells)
            orders[f"oct+cell{cells}"] = np.lexsort((morton, octant))
            orders[f"cell{cells}+oct"] = np.lexsort((octant, morton))
        octant, morton = sort_keys(rays, lo, hi, 16, dirbits=4)
        orders["dir96+cell16"] = np.lexsort((morton, octant))
        orders["cell16+dir96"] = np.lexsort((octant, morton))
        for label, order in orders.items():
            ms, h = timed(ctx, rays[order])
            back = np.empty_like(h)
            back[order] = h
            assert back.tobytes() == hits.tobytes(), "hit records must not depend on the order"
            print(f"gen{g} bounce rays   {label:14s} {ms:8.3f} ms {n / ms / 1e3:9.1f} Mrays/s  x{ms_tile / ms:.2f}", flush=True)
    # the mix a wavefront of the fused kernel sees: every slot holds a ray of SOME generation
    if len(gens) >= 2:
        n = len(cam)
        mix = cam.copy()
        pick = rng.integers(0, len(gens) + 1, n)
        # generation g's ray of slot s exists only if the chain of hits reached it: fall back to the camera ray
        idx = np.arange(n)
        cur_idx = idx
        chain_rays, chain_hits = cam, h0
        for g, rays, _, hits in gens:
            m = chain_hits["hit"] != 0
            cur_idx = cur_idx[m]
            sel = pick[cur_idx] >= g
            mix[cur_idx[sel]] = rays[sel]
            chain_rays, chain_hits = rays, hits
        lo, hi = mix["o"].min(0), mix["o"].max(0)
        ms_tile, hm = timed(ctx, mix)
        print(f"mix of generations tile   {ms_tile:8.3f} ms {n / ms_tile / 1e3:9.1f} Mrays/s  x1.00")
        for cells in (16, 32):
            octant, morton = sort_keys(mix, lo, hi, cells)
            for label, order in ((f"oct+cell{cells}", np.lexsort((morton, octant))), (f"cell{cells}+oct", np.lexsort((octant, morton)))):
                ms, h = timed(ctx, mix[order])
                print(f"mix of generations {label:14s} {ms:8.3f} ms {n / ms / 1e3:9.1f} Mrays/s  x{ms_tile / ms:.2f}", flush=True)


if __name__ == "__main__":
    main()
