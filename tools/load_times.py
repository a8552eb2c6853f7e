#!/usr/bin/env python
"""Scene files -> flat pools: ythip_scene_open / ythip_scene_read against the reference's load_scene (+ the flatten step an
application needs before it can upload), on the same files, pools compared byte for byte.  With a GPU also files -> HBM:
ythip_load_scene (pinned staging pools, one upload) against load_scene + flatten + ythip_upload_scene.

    python tools/load_times.py [--steps 1000] [--env 4096] [--tex 2048] [--shapes 64]

The scene: `shapes` copies of a steps x steps/2 triangulated plane (one PLY each, as the reference writes them), a lat-long
float environment (HDR file, env x env/2) and a byte texture (PNG, tex x tex), written by the reference's save_scene.
Needs oracle/_ref."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch  # before libythip: both must end up on the one libamdhip64 torch ships (two HIP runtimes in a process: the second sees no device)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import parity as P  # noqa: E402
from parity import ry, yt, ysc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--env", type=int, default=4096)
    ap.add_argument("--tex", type=int, default=2048)
    ap.add_argument("--shapes", type=int, default=64)
    ap.add_argument("--repeat", type=int, default=3)
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    sc = ysc.plane_scene(steps=(a.steps, a.steps // 2))
    env = np.kron(rng.random((a.env // 32, a.env // 16, 4)).astype("f4"), np.ones((16, 16, 1), "f4"))
    tex = (np.kron(rng.random((a.tex // 8, a.tex // 8, 4)), np.ones((8, 8, 1))) * 255).astype("u1")
    sc.add_environment(emission=(1, 1, 1), emission_tex=sc.add_texture(env, linear=True))
    sc.materials["color_tex"][0] = sc.add_texture(tex)
    ref = ry.RefScene.from_flat(sc)
    d = tempfile.mkdtemp(prefix="ythip_load_")
    path = os.path.join(d, "scene.json")
    ref.save(path)
    if a.shapes > 1:  # the same shape file under many names: many-file behaviour without many GB of authoring
        import json
        doc = json.load(open(path))
        first = doc["shapes"][0]
        for k in range(1, a.shapes):
            uri = f"shapes/copy{k}.ply"
            os.link(os.path.join(d, first["uri"]), os.path.join(d, uri))
            doc["shapes"].append({"name": f"copy{k}", "uri": uri})
            doc["instances"].append({"name": f"copy{k}", "shape": len(doc["shapes"]) - 1, "material": 0,
                                     "frame": [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0.01 * k, 0]})
        json.dump(doc, open(path, "w"))
    size = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(d) for f in set(fs)) / 1e6
    print(f"scene: {a.shapes} x {a.steps}x{a.steps // 2} plane PLYs, HDR {a.env}x{a.env // 2}, PNG {a.tex}x{a.tex}: "
          f"{size:.0f} MB of files (hard links counted once per name), host threads {os.cpu_count()}")

    def best(fn):
        ts, out = [], None
        for _ in range(a.repeat):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return min(ts), out

    t_ref, theirs = best(lambda: ry.RefScene.load(path))
    t_flat, flat = best(lambda: theirs.flat())
    t_ours, (mine, _, _) = best(lambda: yt.load_scene_file(path))
    t_one, _ = best(lambda: yt.load_scene_file(path, threads=1))
    same = all(np.ascontiguousarray(getattr(mine, f)).tobytes() == np.ascontiguousarray(getattr(flat, f)).tobytes() for f in P._FIELDS)
    pool_mb = sum(np.asarray(getattr(flat, f)).nbytes for f in P._FIELDS) / 1e6
    print(f"pools: {pool_mb:.0f} MB, identical: {same}")
    print(f"reference load_scene            {t_ref * 1e3:9.1f} ms   (+ flatten {t_flat * 1e3:.1f} ms = {(t_ref + t_flat) * 1e3:.1f} ms)")
    print(f"ythip_scene_open + _read        {t_ours * 1e3:9.1f} ms   ({(t_ref + t_flat) / t_ours:.2f}x; one thread {t_one * 1e3:.1f} ms)")
    if torch.cuda.is_available():
        ctx = yt.Context(0)
        ctx.load_scene(path)  # first use: staging pools are allocated (hipHostMalloc), excluded like the reference's first malloc

        def ordinary():
            s = ry.RefScene.load(path).flat()
            ctx.upload_scene(s)
            ctx.sync()
        t_gpu, _ = best(lambda: (ctx.lib.ythip_load_scene(ctx.h, path.encode(), 0, None), ctx.sync()))
        t_ord, _ = best(ordinary)
        print(f"files -> HBM: ythip_load_scene   {t_gpu * 1e3:9.1f} ms")
        print(f"files -> HBM: load_scene + flatten + ythip_upload_scene {t_ord * 1e3:9.1f} ms   ({t_ord / t_gpu:.2f}x)")
        ctx.close()
    import shutil
    shutil.rmtree(d)
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
