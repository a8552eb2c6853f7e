"""BASELINE configs[4]'s geometry as a fixture.

The scene is `make_hair(make_sphere(32, 1), {8, 100000}, {0.2, 0.2}, {0.002, 0.001})` (SURVEY.md §8d;
yocto_shape.cpp:1264-1334).  Its only compiler-dependent part is sample_shape's stream (unsequenced rng arguments,
SURVEY.md Appendix A-13): the 100,000 strand roots and their normals.  This script asks the g++-built reference
(oracle/_ref) for the hair ONCE and stores what cannot be recomputed — root position and normal per strand, 2.4 MB —
in tests/golden/hair_roots.npz; everything else (strand lengths from the reference's own PCG stream, the 900,000
vertices, radii, the 800,000 segments) is restated in numpy by scenes.hair_scene() and checked against the
reference's arrays bit for bit, here and in tests/test_host.py.

Run:  python tests/golden/make_hair_fixture.py        (needs /root/reference -> oracle/_ref)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import parity as P  # noqa: E402
from parity import ry, ysc  # noqa: E402


def reference_hair_scene():
    sys.path.insert(0, os.path.dirname(HERE))
    import test_gpu_baseline_configs as T
    return T.hair_scene()


def main():
    assert P.have_ref(), "oracle/_ref is needed (run python -c 'import __graft_entry__ as g; g.build()')"
    flat = reference_hair_scene()
    hair = flat.shape_arrays(1)
    steps = 8
    pos, nrm = hair["positions"].reshape(-1, steps + 1, 3), hair["normals"].reshape(-1, steps + 1, 3)
    roots, normals = pos[:, 0, :].copy(), nrm[:, 0, :].copy()
    out = os.path.join(HERE, "hair_roots.npz")
    base = {"base_" + k: v for k, v in flat.shape_arrays(0).items() if v is not None and len(v)}
    np.savez_compressed(out, roots=roots, normals=normals, **base)
    print("wrote", out, os.path.getsize(out), "bytes")
    mine = ysc.hair_scene()
    for k in ["cameras", "instances", "environments", "shapes", "materials"] + [p[0] for p in type(flat).POOLS]:
        a, b = getattr(flat, k), getattr(mine, k)
        assert a.tobytes() == b.tobytes(), f"scenes.hair_scene() differs from the reference's scene in {k}"
    print("scenes.hair_scene() == the reference's make_hair scene, every pool, byte for byte")


if __name__ == "__main__":
    main()
