#!/usr/bin/env python
"""Random scenes x random trace_params: whole trace_state of the device render against the
live reference (oracle/_ref), byte for byte.  TEST INFRASTRUCTURE (the checker is oracle/_ref).
  python tools/fuzz_parity.py [first_seed] [count]        prints one line per case, FAIL lines name the arrays
Every feature the path has is drawn at random: the four primitive kinds (with / without
normals, texcoords, colors, radius), all eight material types incl. delta lobes, opacity,
emission (area lights), every texture slot (byte / float, nearest, clamp), constant and
textured environments in rotated frames, non-rigid instance frames, perspective / orthographic
cameras with and without aperture, all nine samplers and the falsecolor modes, bounces 0-8,
clamp, nocaustics, envhidden, tentfilter, progressive batches, split_middle / split_sah trees."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity as P
yt, ysc, ry = P.yt, P.ysc, P.ry
f32 = np.float32


def random_frame(r, spread=1.5, nonrigid=0.5):
    a = r.normal(size=(3, 3))
    q, _ = np.linalg.qr(a)
    if r.random() < nonrigid:
        q = q * r.uniform(0.5, 1.6, 3)[None, :]
        if r.random() < 0.3:
            q[0] += 0.3 * q[1]  # shear
    fr = np.zeros(12, f32)
    fr[0:9] = q.T.reshape(9)
    fr[9:12] = r.uniform(-spread, spread, 3)
    return fr


def random_texture(r, sc):
    h, w = int(r.integers(1, 20)), int(r.integers(1, 20))
    px = r.random((h, w, 4)).astype(f32)
    if r.random() < 0.5:
        return sc.add_texture((px * 255).astype(np.uint8), linear=bool(r.random() < 0.3), nearest=bool(r.random() < 0.3),
                              clamp=bool(r.random() < 0.3))
    return sc.add_texture(px * f32(r.uniform(0.5, 2.0)), linear=bool(r.random() < 0.7), nearest=bool(r.random() < 0.3),
                          clamp=bool(r.random() < 0.3))


def random_shape(r, sc):
    kind = r.choice(["triangles", "quads", "lines", "points", "sphere", "plane"])
    if kind == "sphere":
        s = ysc.make_uvsphere((int(r.integers(3, 12)), int(r.integers(2, 8))), float(r.uniform(0.2, 0.7)))
        if r.random() < 0.5:
            s = ysc.triangulated(s)
        if r.random() < 0.3:
            s.pop("normals", None)
        if r.random() < 0.3:
            s.pop("texcoords", None)
        return ysc.add_shape(sc, s)
    if kind == "plane":
        s = ysc.make_recty((int(r.integers(1, 6)), int(r.integers(1, 6))), (float(r.uniform(0.5, 3)), float(r.uniform(0.5, 3))))
        if r.random() < 0.5:
            s = ysc.triangulated(s)
        return ysc.add_shape(sc, s)
    n = int(r.integers(1, 60))
    c = r.uniform(-1, 1, (n, 3)).astype(f32)
    kw = {}
    if kind == "triangles":
        pos = (c[:, None, :] + r.normal(size=(n, 3, 3)) * 0.3).reshape(-1, 3).astype(f32)
        kw["triangles"] = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
    elif kind == "quads":
        pos = (c[:, None, :] + r.normal(size=(n, 4, 3)) * 0.3).reshape(-1, 3).astype(f32)
        q = np.arange(4 * n, dtype=np.int32).reshape(-1, 4)
        q[::3, 3] = q[::3, 2]
        kw["quads"] = q
    elif kind == "lines":
        pos = (c[:, None, :] + r.normal(size=(n, 2, 3)) * 0.4).reshape(-1, 3).astype(f32)
        kw["lines"] = np.arange(2 * n, dtype=np.int32).reshape(-1, 2)
        kw["radius"] = r.uniform(0.005, 0.08, len(pos)).astype(f32)
    else:
        pos = c
        kw["points"] = np.arange(n, dtype=np.int32)
        kw["radius"] = r.uniform(0.02, 0.15, n).astype(f32)
    if r.random() < 0.5:
        nr = r.normal(size=pos.shape)
        kw["normals"] = (nr / np.linalg.norm(nr, axis=1, keepdims=True)).astype(f32)
    if r.random() < 0.5:
        kw["texcoords"] = r.uniform(-0.5, 1.5, (len(pos), 2)).astype(f32)
    if r.random() < 0.3:
        kw["colors"] = r.uniform(0, 1, (len(pos), 4)).astype(f32)
    return sc.add_shape(pos, **kw)


def random_material(r, sc, textures):
    t = lambda p=0.3: int(r.choice(textures)) if textures and r.random() < p else -1
    kind = yt.MATERIAL_TYPES[int(r.integers(0, 8))]
    kw = dict(type=kind, color=tuple(r.uniform(0.05, 1, 3)), roughness=float(r.choice([0.0, r.uniform(0.02, 1)])),
              metallic=float(r.choice([0.0, r.uniform(0, 1)])), ior=float(r.uniform(1.1, 2.2)),
              scattering=tuple(r.uniform(0, 1, 3)), scanisotropy=float(r.uniform(-0.7, 0.7)),
              trdepth=float(r.uniform(0.01, 1.0)), opacity=float(r.choice([1.0, 1.0, r.uniform(0.2, 0.95)])),
              color_tex=t(), roughness_tex=t(0.2), scattering_tex=t(0.2), normal_tex=t(0.2), emission_tex=t(0.2))
    if r.random() < 0.2:
        kw["emission"] = tuple(r.uniform(0, 8, 3))
    return sc.add_material(**kw)


def random_scene(seed):
    r = np.random.default_rng(seed)
    sc = yt.FlatScene()
    eye = r.uniform(-1, 1, 3) + np.array([0, 0.5, 4.0])
    sc.add_camera(ysc.lookat_frame(tuple(eye), tuple(r.uniform(-0.5, 0.5, 3))), lens=float(r.uniform(0.02, 0.1)),
                  film=0.036, aspect=float(r.choice([1.0, 1.5, 0.75, 16 / 9])), focus=float(r.uniform(2, 6)),
                  aperture=float(r.choice([0.0, 0.0, r.uniform(0.01, 0.2)])), orthographic=bool(r.random() < 0.15))
    textures = [random_texture(r, sc) for _ in range(int(r.integers(0, 5)))]
    shapes = [random_shape(r, sc) for _ in range(int(r.integers(1, 6)))]
    materials = [random_material(r, sc, textures) for _ in range(int(r.integers(1, 7)))]
    for _ in range(int(r.integers(1, 12))):
        fr = yt.IDENTITY_FRAME.copy() if r.random() < 0.3 else random_frame(r)
        sc.add_instance(int(r.choice(shapes)), int(r.choice(materials)), fr)
    if r.random() < 0.5:  # a guaranteed area light
        s = ysc.add_shape(sc, ysc.make_recty((1, 1), (0.6, 0.6)))
        m = sc.add_material("matte", emission=tuple(r.uniform(2, 12, 3)))
        sc.add_instance(s, m, random_frame(r, 2.0, 0.3))
    ne = int(r.choice([0, 1, 1, 2]))
    # (the reference reads out of bounds in sample_lights when a scene has no light at all: make_trace_lights
    # takes instances whose material emits and whose shape has triangles or quads, yocto_trace.cpp:1528-1560)
    lit = any(np.any(sc.materials["emission"][i["material"]] > 0) and
              (sc.shapes[i["shape"]]["num_triangles"] > 0 or sc.shapes[i["shape"]]["num_quads"] > 0) for i in sc.instances)
    if ne == 0 and not lit:
        ne = 1
    for _ in range(ne):
        sc.add_environment(tuple(r.uniform(0.1, 1.5, 3)), emission_tex=int(r.choice(textures)) if textures and r.random() < 0.5 else -1,
                           frame=yt.IDENTITY_FRAME.copy() if r.random() < 0.5 else random_frame(r, 0.0, 0.0))
    sampler = yt.SAMPLERS[int(r.integers(0, 9))]
    if sampler == "diagram":
        sampler = "path"
    spp = int(r.integers(1, 5))
    p = dict(sampler=sampler, resolution=int(r.choice([48, 64, 97, 130])), samples=spp, batch=int(r.choice([1, spp])),
             bounces=int(r.choice([0, 1, 2, 4, 8])), clamp=float(r.choice([10.0, 1.0, 100.0])),
             nocaustics=int(r.random() < 0.3), envhidden=int(r.random() < 0.3), tentfilter=int(r.random() < 0.3),
             seed=int(r.integers(1, 1 << 30)), falsecolor=int(r.integers(0, 18)))
    # round 6: some `path` cases run on the streaming scheduler (csrc/yt_stream.h) — wide walk forced (the scheduler has no binary
    # walk), batches of >= 4 samples, random sort order / cell grid / number of chains; drawn AFTER everything else, so the scenes
    # and parameters of a seed are what they were in rounds 2-5
    stream = None
    if sampler in ("path", "pathdirect", "naive", "pathtest") and p["bounces"] > 0 and r.random() < 0.6:  # (pathdirect: since the round's last session)
        spp = int(r.integers(4, 9))
        p["samples"], p["batch"] = spp * int(r.integers(1, 3)), spp
        stream = dict(order=int(r.integers(0, 3)), cell_bits=int(r.integers(1, 6)), groups=int(r.choice([1, 2, 3])),
                      finish=int(r.choice([0, 100, 250, 500, 800, 1000])), phased=int(r.integers(0, 2)))  # (round 6, last session: the tail kernel)
    return sc, p, bool(r.random() < 0.25), stream


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    fails = 0
    t0 = time.time()
    for seed in range(first, first + count):
        flat, pk, hq, stream = random_scene(seed)
        p = yt.trace_params(**pk)
        ran = ""
        try:
            ctx = P.gpu_context(flat, highquality=hq)
            if stream:
                os.environ["YTHIP_STREAM_MIN_SLOTS"] = "1024"  # (small frames: let two or three chains form)
                ctx.set_traversal("wide")
                ctx.set_scheduler(1)
                ctx.set_stream_options(order=stream["order"], cell_bits=stream["cell_bits"])
                ctx.set_stream_groups(stream["groups"])
                ctx.set_stream_finish(stream["finish"])
                ctx.set_stream_options(phased=stream["phased"])
            gpu = P.gpu_render(ctx, flat, p)
            if stream:
                info = ctx.stream_info()
                ran = f" streamed {info['ran']} ({info['generations']} generations, {info['groups']} chains, order {stream['order']}, {stream['cell_bits']} cell bits, tail kernel at {stream['finish']}: {info['finish_rays']} rays, phased {stream['phased']})"
            ctx.close()
            ref = P.RefBundle(flat, highquality=hq).render(p)
        except Exception as e:  # a scene one side refuses: report, go on
            print(f"seed {seed}: ERROR {type(e).__name__}: {str(e)[:200]}", flush=True)
            fails += 1
            continue
        bad = [k for k in ["image", "albedo", "normal", "hits", "rngs"]
               if np.asarray(gpu[k]).tobytes() != np.asarray(ref[k]).tobytes()]
        tag = "ok  " if not bad else "FAIL"
        if bad:
            fails += 1
        extra = ""
        if bad:
            a, b = np.asarray(gpu["image"]).reshape(-1, 4), np.asarray(ref["image"]).reshape(-1, 4)
            nd = int(np.sum(np.any(a.view("u4") != b.view("u4"), axis=1)))
            extra = f" differing: {bad}, image pixels {nd}/{len(a)}"
        print(f"seed {seed}: {tag} {pk['sampler']:10s} res {pk['resolution']:3d} spp {pk['samples']} bounces {pk['bounces']} "
              f"hq {int(hq)} shapes {len(flat.shapes)} inst {len(flat.instances)} mats {len(flat.materials)} "
              f"tex {len(flat.textures)} env {len(flat.environments)}{ran}{extra}", flush=True)
    print(f"{count} cases, {fails} failures, {time.time() - t0:.0f} s")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
