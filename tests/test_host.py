"""CPU-side tests (`-m "not gpu"`): the oracle against the reference's known
answers, the host logic (BVH builder, lights, rng seeding, scene generators)
against the compiled reference, and that libythip.so loads and exports every
symbol of include/ythip.h.  No GPU compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import parity as P
from parity import ry, yt

ROOT = P.ROOT
needs_ref = pytest.mark.skipif(not P.have_ref(), reason="oracle/_ref not built")
KAT = np.load(os.path.join(P.GOLDEN, "kat.npz"))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ythip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)  # strip comments
    declared = set(re.findall(r"\b(ythip_[a-z_0-9]+)\s*\(", header))
    lib = yt.load_library()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in ythip.h but not exported"
    assert declared == set(yt.exported_symbols()), declared ^ set(yt.exported_symbols())


def test_header_is_plain_c():
    """The boundary is a C ABI: include/ythip.h compiles as C99 on its own (plain pointers
    and sizes, no C++ or torch types), and a C translation unit can call through it."""
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    inc = os.path.join(ROOT, "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                    os.path.join(inc, "ythip.h")], check=True)
    src = """#include <stdio.h>
#include "ythip.h"
int main(void) {
  ythip_ctx* ctx = 0;
  int rc = ythip_create(0, &ctx);           /* no GPU here: must fail loudly, not fall back */
  printf("%d %s\\n", rc, ythip_last_error(ctx));
  if (ctx) ythip_destroy(ctx);
  return 0;
}
"""
    libdir = os.path.join(ROOT, "yocto-gl_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "abi.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "abi")
        subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", inc, c, "-o", exe, "-L", libdir, "-lythip",
                        "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        rc, _, msg = r.stdout.strip().partition(" ")
        import torch
        if not torch.cuda.is_available():
            assert int(rc) != 0 and "no CPU fallback" in msg, r.stdout


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(yt.YthipError):
        yt.Context(0)


def test_pod_struct_sizes_match_reference_layouts():
    # SURVEY.md §8a sizes
    assert yt.camera_dt.itemsize == 72 and yt.instance_dt.itemsize == 56
    assert yt.material_dt.itemsize == 84 and yt.environment_dt.itemsize == 64
    assert yt.node_dt.itemsize == 32 and yt.hit_dt.itemsize == 24 and yt.ray_dt.itemsize == 32
    assert C.sizeof(yt.CParams) == 80  # trace_params (72 B) + this library's `fastmath` word, padded


def test_pcg_seeding_known_answers():
    # make_trace_state(make_cornellbox(), res=256)
    r = yt.make_rngs(yt.TRACE_DEFAULT_SEED, 256 * 256)
    assert tuple(r[0]) == (10473825460370060855, 1450249601)
    assert tuple(r[-1]) == (1965839424625278335, 94108541)
    assert np.array_equal(r[0], KAT["rngs256_first"]) and np.array_equal(r[-1], KAT["rngs256_last"])


def test_state_size_rule():
    cam = np.zeros(1, yt.camera_dt)
    cam["aspect"] = 16 / 9
    assert yt.state_size(cam[0], 1280) == (1280, 720)
    cam["aspect"] = 1.0
    assert yt.state_size(cam[0], 256) == (256, 256)
    cam["aspect"] = 0.5
    assert yt.state_size(cam[0], 256) == (128, 256)


@needs_ref
def test_oracle_known_answers_from_survey():
    """Pins the compiled reference: PCG, hit-index hash, image/rng hashes
    (SURVEY.md §8c, Appendix A matrix)."""
    assert tuple(ry.make_rng(961748941, 1)) == (17286221497386715027, 3)
    r = ry.rand1f(ry.make_rng(961748941, 1), 4)[0]
    assert np.allclose(r, [0.893633127, 0.246839881, 0.458433747, 0.477094531], atol=1e-9)
    sc = ry.RefScene.cornellbox()
    p = yt.trace_params(resolution=256, sampler="falsecolor", samples=1)
    st = ry.RefState(sc, p)
    bvh = ry.RefBvh(sc)
    rays = ry.camera_rays(st, sc, p)
    hits = ry.intersect_batch(bvh, sc, rays)
    assert hits["hit"].sum() == 65536
    pairs = np.stack([hits["instance"], hits["element"]], 1).astype("<i4")
    assert ry.fnv1a64(pairs.tobytes()) == 0xE32777791F766A25
    assert abs(hits["distance"].astype("f8").sum() - 264850.978779) < 1e-3
    p2 = yt.trace_params(resolution=128, sampler="path", samples=8, batch=8, noparallel=1)
    st2 = ry.RefState(sc, p2)
    ry.trace_samples(st2, sc, bvh, ry.RefLights(sc), p2)
    s2 = st2.get()
    assert ry.fnv1a64(s2["image"].tobytes()) == 0xA2D4ACB1A35EE25C
    assert ry.fnv1a64(s2["rngs"].tobytes()) == 0x57E409C41EF42565


@needs_ref
def test_camera_ray_convention_matches_falsecolor_render():
    """The luv-before-puv draw order (g++), pinned through the reference's own
    trace_samples: falsecolor=position of the reference == eval of our ray batch."""
    sc = ry.RefScene.cornellbox()
    p = yt.trace_params(resolution=64, sampler="falsecolor", falsecolor="position", samples=1)
    st = ry.RefState(sc, p)
    bvh, lights = ry.RefBvh(sc), ry.RefLights(sc)
    rays = ry.camera_rays(st, sc, p)
    hits = ry.intersect_batch(bvh, sc, rays)
    sh = ry.eval_shading(sc, hits, -rays["d"])
    ry.trace_samples(st, sc, bvh, lights, p)
    img = st.get()["image"]
    pos = sh[:, :3] * np.float32(0.5) + np.float32(0.5)
    lin = np.where(pos <= 0.04045, pos / 12.92, ((pos + 0.055) / 1.055) ** 2.4)
    assert np.allclose(img[:, :3], lin, rtol=2e-6, atol=1e-7)


@needs_ref
@pytest.mark.parametrize("name", list(P.SCENES))
@pytest.mark.parametrize("hq", [False, True])
def test_host_bvh_builder_is_node_for_node_identical(name, hq):
    flat = P.SCENES[name]()
    mine = yt.host_make_bvh(flat, hq)
    ref = ry.RefBvh(ry.RefScene.from_flat(flat), hq).flat()
    assert np.array_equal(mine.node_offset, ref.node_offset)
    assert np.array_equal(mine.primitives, ref.primitives)
    assert mine.nodes.tobytes() == ref.nodes.tobytes()


@needs_ref
@pytest.mark.parametrize("name", list(P.SCENES))
def test_host_lights_builder_identical(name):
    flat = P.SCENES[name]()
    mine = yt.host_make_lights(flat)
    ref = ry.RefLights(ry.RefScene.from_flat(flat)).flat()
    assert mine.lights.tobytes() == ref.lights.tobytes()
    assert mine.cdf.tobytes() == ref.cdf.tobytes()


@needs_ref
def test_plane_generator_matches_reference_generators():
    """cfg2 scene: our numpy make_recty + quads_to_triangles == the reference's."""
    mine = P.ysc.plane_scene(steps=(200, 100))
    rs = ry.RefScene.new()
    sh = rs.add_recty((200, 100), (10, 10))
    m = rs.add_material(type="matte", color=(0.7, 0.7, 0.7))
    rs.add_instance(sh, m)
    rs.add_environment((1, 1, 1))
    rs.add_camera_lookat((0, 3, 8), (0, 0, 0), lens=0.035, film=0.036, aspect=16 / 9)
    ref = rs.flat()
    for k in ["positions", "normals", "texcoords", "triangles"]:
        assert getattr(mine, k).tobytes() == getattr(ref, k).tobytes(), k
    assert mine.cameras.tobytes() == ref.cameras.tobytes()
    assert mine.instances.tobytes() == ref.instances.tobytes()
    assert mine.materials.tobytes() == ref.materials.tobytes()
    assert mine.environments.tobytes() == ref.environments.tobytes()


def test_hair_scene_is_configs4_as_specified():
    """bench.py's configs[4] = scenes.hair_scene(): 100,000 strands x 8 segments with radii over make_sphere(32, 1)
    (SURVEY.md §8d), from the committed strand roots (tests/golden/hair_roots.npz)."""
    flat = P.ysc.hair_scene()
    assert len(flat.shapes) == 2 and int(flat.shapes[1]["num_lines"]) == 800_000
    assert int(flat.shapes[1]["num_positions"]) == 900_000 and int(flat.shapes[1]["num_radius"]) == 900_000
    assert int(flat.shapes[0]["num_quads"]) == 6 * 32 * 32
    r = flat.shape_arrays(1)["radius"].reshape(-1, 9)
    assert (r[:, 0] == np.float32(0.002)).all() and (r[:, 8] == np.float32(0.001)).all()


@needs_ref
def test_hair_scene_equals_the_reference_make_hair_scene():
    """... and it IS the scene the full-size parity test renders (tests/test_gpu_baseline_configs.py builds it through
    the g++ reference's make_hair): every pool, byte for byte.  The numpy restatement of make_hair / make_lines / the
    strand-length PCG stream (scenes.hair_scene) is therefore the reference's (yocto_shape.cpp:1264-1334, :962-990)."""
    import test_gpu_baseline_configs as T
    ref, mine = T.hair_scene(), P.ysc.hair_scene()
    for k in ["cameras", "instances", "environments", "shapes", "materials"] + [p[0] for p in type(ref).POOLS]:
        assert getattr(ref, k).tobytes() == getattr(mine, k).tobytes(), k


def test_full_size_plane_bvh_shape():
    """BASELINE config 2: 1,000,000 triangles → 503,478 nodes."""
    flat = P.ysc.plane_scene()
    assert int(flat.shapes[0]["num_triangles"]) == 1_000_000 and len(flat.positions) == 501_501
    bvh = yt.host_make_bvh(flat)
    assert int(bvh.node_offset[-1]) == 503_478  # 503,477 BLAS + 1 TLAS node
    prims = bvh.primitives[: int(bvh.prim_offset[1])]
    assert np.array_equal(np.sort(prims), np.arange(1_000_000))  # a permutation


@needs_ref
def test_golden_fixtures_are_reproducible():
    """The committed fixtures are what the compiled reference produces today."""
    flat = P.SCENES["cornellbox"]()
    rb = P.RefBundle(flat)
    g = np.load(os.path.join(P.GOLDEN, "render_cornellbox_path.npz"))
    st = rb.render(yt.trace_params(sampler="path", resolution=32, samples=4, batch=4,
                                   falsecolor="normal"))
    assert st["image"].tobytes() == g["image"].tobytes()
    assert st["rngs"].tobytes() == g["rngs"].tobytes()
    gh = np.load(os.path.join(P.GOLDEN, "hits_cornellbox.npz"))
    h = ry.intersect_batch(rb.bvh, rb.scene, P.random_rays(flat, 2048))
    assert P.hits_equal(h, gh["hits"])
