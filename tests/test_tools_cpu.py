"""CPU checks of the round-2 helpers that run around the GPU path (no device needed)."""
import os
import sys

import numpy as np

import parity as P

ROOT = P.ROOT
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd"))


def test_fuzz_scenes_are_reproducible_and_always_lit():
    """tools/fuzz_parity.py: a seed names one scene + one parameter set, and every scene has a
    light the reference's make_trace_lights would find (the reference reads out of bounds in
    sample_lights otherwise)."""
    import fuzz_parity as F
    samplers = set()
    for seed in range(60):
        a, pa, ha = F.random_scene(seed)
        b, pb, hb = F.random_scene(seed)
        assert pa == pb and ha == hb
        assert a.positions.tobytes() == b.positions.tobytes() and a.materials.tobytes() == b.materials.tobytes()
        samplers.add(pa["sampler"])
        lit = len(a.environments) > 0 or any(
            np.any(a.materials["emission"][i["material"]] > 0) and
            (a.shapes[i["shape"]]["num_triangles"] > 0 or a.shapes[i["shape"]]["num_quads"] > 0) for i in a.instances)
        assert lit, seed
    assert len(samplers) >= 6
