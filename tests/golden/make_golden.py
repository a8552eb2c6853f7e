"""Generates the committed golden fixtures from the COMPILED REFERENCE
(oracle/_ref/libyocto_ref.so, built from /root/reference by oracle/Makefile).
Run here (where /root/reference exists):  python tests/golden/make_golden.py

  cornellbox.npz            make_cornellbox() as a flat scene
  hits_<scene>.npz          intersect_scene_bvh on a seeded ray batch + instance walks
  render_<scene>_<sampler>.npz   trace_samples at 32 px, default seed
  kat.npz                   PCG / seeding / hash known answers
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import parity as P  # noqa: E402
from parity import ry, yt  # noqa: E402

GOLDEN_RENDERS = {
    "cornellbox": ["path", "pathdirect", "pathmis", "pathtest", "naive", "eyelight",
                   "diagram", "furnace", "falsecolor"],
    "materials": ["path", "pathdirect", "pathmis", "naive", "eyelight", "falsecolor"],
    "plane": ["path", "eyelight"],
    "instances": ["path", "eyelight", "falsecolor"],
    "lines_points": ["path", "eyelight", "falsecolor"],
}
RES, SPP = 32, 4


def main():
    assert ry.available(), "build oracle/_ref first (make -C oracle ref)"
    # 1. cornell box exported from the reference
    P.save_scene(os.path.join(HERE, "cornellbox.npz"), ry.RefScene.cornellbox().flat())
    # 2. KATs
    sc = ry.RefScene.cornellbox()
    p = yt.trace_params(resolution=256, sampler="falsecolor", samples=1)
    st = ry.RefState(sc, p)
    bvh = ry.RefBvh(sc)
    lights = ry.RefLights(sc)
    rays = ry.camera_rays(st, sc, p)
    hits = ry.intersect_batch(bvh, sc, rays)
    pairs = np.stack([hits["instance"], hits["element"]], 1).astype("<i4")
    p2 = yt.trace_params(resolution=128, sampler="path", samples=8, batch=8, noparallel=1)
    st2 = ry.RefState(sc, p2)
    ry.trace_samples(st2, sc, bvh, lights, p2)
    s2 = st2.get()
    np.savez(os.path.join(HERE, "kat.npz"),
             make_rng=ry.make_rng(961748941, 1),
             rand1f=ry.rand1f(ry.make_rng(961748941, 1), 8)[0],
             rngs256_first=st.get()["rngs"][0], rngs256_last=st.get()["rngs"][-1],
             hit_hash=np.uint64(ry.fnv1a64(pairs.tobytes())),
             hit_dist_sum=hits["distance"].astype("f8").sum(),
             image_hash=np.uint64(ry.fnv1a64(s2["image"].tobytes())),
             rngs_hash=np.uint64(ry.fnv1a64(s2["rngs"].tobytes())))
    # 3. per-scene hits + renders
    for name, fn in P.SCENES.items():
        flat = fn()
        rb = P.RefBundle(flat)
        rays = P.random_rays(flat, 2048)
        h = ry.intersect_batch(rb.bvh, rb.scene, rays)
        h_any = ry.intersect_batch(rb.bvh, rb.scene, rays, find_any=True)
        inst = (np.arange(len(rays)) % len(flat.instances)).astype("i4")
        hi = ry.intersect_instance_batch(rb.bvh, rb.scene, inst, rays)
        np.savez_compressed(os.path.join(HERE, f"hits_{name}.npz"), hits=h, hits_any=h_any,
                            inst=inst, hits_inst=hi)
        for sampler in GOLDEN_RENDERS[name]:
            params = yt.trace_params(sampler=sampler, resolution=RES, samples=SPP, batch=SPP,
                                     falsecolor="normal")
            st = rb.render(params)
            np.savez_compressed(os.path.join(HERE, f"render_{name}_{sampler}.npz"),
                                image=st["image"], albedo=st["albedo"], normal=st["normal"],
                                hits=st["hits"], rngs=st["rngs"])
        print(name, "done", flush=True)


if __name__ == "__main__":
    main()
