"""On-device make_bvh (yt_gpubuild.hip), `-m gpu`.

The bar is the same as for the host builder (tests/test_host.py): the tree the
GPU builds is the reference's tree NODE FOR NODE AND BIT FOR BIT — node order,
`primitives` permutation, boxes — checked against the host builder (itself pinned
against the live reference's make_scene_bvh), and against the live reference
directly when it travelled.  The baked traversal arrays must be byte-identical
under both builders, so every hit record and image is too."""
import time

import numpy as np
import pytest

import parity as P
from parity import ry, yt, ysc

pytestmark = pytest.mark.gpu
f32 = np.float32


def build_both(flat, min_prims=8, highquality=False):
    """(device-built, host-built) downloads + baked arrays of the same scene."""
    out = []
    for mode in ["device", "host"]:
        ctx = yt.Context(0)
        ctx.upload_scene(flat)
        ctx.set_bvh_builder(mode, min_prims)
        ctx.make_trace_bvh(flat, highquality)
        out.append((ctx.download_bvh(), ctx.download_baked_bvh(), ctx.bvh_build_info()))
        ctx.close()
    return out


def assert_same(dev, host, what=""):
    (db, (dp, dl, dq), di), (hb, (hp, hl, hq), hi) = dev, host
    assert np.array_equal(db.node_offset, hb.node_offset), what
    assert np.array_equal(db.prim_offset, hb.prim_offset), what
    assert db.primitives.tobytes() == hb.primitives.tobytes(), what + " primitives"
    assert db.nodes.tobytes() == hb.nodes.tobytes(), what + " nodes"
    assert dp.tobytes() == hp.tobytes(), what + " baked pairs"
    assert dl.tobytes() == hl.tobytes(), what + " baked leaf data"
    assert dq.tobytes() == hq.tobytes(), what + " baked quads"


@pytest.mark.parametrize("name", list(P.SCENES))
def test_device_tree_equals_host_tree_on_test_scenes(name):
    flat = P.SCENES[name]()
    dev, host = build_both(flat, min_prims=5)  # force even the tiny shapes through the GPU
    assert dev[2]["device_trees"] >= 1 and host[2]["device_trees"] == 0
    assert_same(dev, host, name)
    if P.have_ref():
        ref = ry.RefBvh(ry.RefScene.from_flat(flat)).flat()
        assert dev[0].nodes.tobytes() == ref.nodes.tobytes()
        assert dev[0].primitives.tobytes() == ref.primitives.tobytes()


def _soup(kind, n, seed, clustered=False, dup=False, grid=False):
    """Random primitive soups that exercise every branch of split_middle."""
    rng = np.random.default_rng(seed)
    sc = yt.FlatScene()
    sc.add_camera(ysc.lookat_frame((0, 0, 5), (0, 0, 0)))
    if grid:  # many identical centroids along axes: ties in `center < split`
        c = rng.integers(0, 8, (n, 3)).astype(f32)
    elif clustered:  # exponentially spaced clusters: very unbalanced middle splits
        c = (rng.normal(size=(n, 3)) * 0.01 + (2.0 ** rng.integers(-6, 6, (n, 1)))).astype(f32)
    else:
        c = (rng.random((n, 3)) * 4 - 2).astype(f32)
    if dup:  # all primitives identical: csize == 0 → "break in half" path
        c[:] = c[0]
    m = sc.add_material("matte", color=(0.5, 0.5, 0.5))
    if kind == "triangles":
        d = (rng.normal(size=(n, 3, 3)) * 0.05).astype(f32)
        if dup:
            d[:] = d[0]
        pos = (c[:, None, :] + d).reshape(-1, 3)
        s = sc.add_shape(pos, triangles=np.arange(3 * n, dtype=np.int32).reshape(-1, 3))
    elif kind == "quads":
        d = (rng.normal(size=(n, 4, 3)) * 0.05).astype(f32)
        pos = (c[:, None, :] + d).reshape(-1, 3)
        q = np.arange(4 * n, dtype=np.int32).reshape(-1, 4)
        q[::7, 3] = q[::7, 2]  # degenerate quads (triangles)
        s = sc.add_shape(pos, quads=q)
    elif kind == "lines":
        d = (rng.normal(size=(n, 2, 3)) * 0.05).astype(f32)
        pos = (c[:, None, :] + d).reshape(-1, 3)
        s = sc.add_shape(pos, lines=np.arange(2 * n, dtype=np.int32).reshape(-1, 2),
                         radius=(rng.random(2 * n) * 0.01 + 0.001).astype(f32))
    else:
        s = sc.add_shape(c, points=np.arange(n, dtype=np.int32),
                         radius=(rng.random(n) * 0.02 + 0.001).astype(f32))
    sc.add_instance(s, m)
    sc.add_environment((1, 1, 1))
    return sc


@pytest.mark.parametrize("kind", ["triangles", "quads", "lines", "points"])
@pytest.mark.parametrize("variant", ["uniform", "clustered", "grid"])
def test_device_tree_equals_host_tree_on_random_soups(kind, variant):
    flat = _soup(kind, 30011, seed=len(kind) * 7 + len(variant),
                 clustered=variant == "clustered", grid=variant == "grid")
    dev, host = build_both(flat)
    assert dev[2]["device_trees"] == 1 and dev[2]["fallbacks"] == 0
    assert_same(dev, host, f"{kind}/{variant}")


def test_identical_primitives_take_the_break_in_half_path():
    dev, host = build_both(_soup("triangles", 4099, seed=3, dup=True))
    assert dev[2]["device_trees"] == 1
    assert_same(dev, host, "dup")


def test_signed_zero_ties_fall_back_to_the_host_builder():
    """A box face at 0 fed by both +0 and -0: the sign the reference keeps depends on
    the serial visiting order → the device builder must notice and hand over."""
    flat = _soup("triangles", 5000, seed=9)
    pos = flat.positions.copy()
    pos[:, 1] = 0.0
    pos[::2, 1] = -0.0
    flat.positions = pos
    dev, host = build_both(flat)
    assert dev[2]["fallbacks"] == 1 and dev[2]["device_trees"] == 0
    assert_same(dev, host, "signed zero")
    # a single sign of zero is no tie: stays on the device
    pos[:, 1] = 0.0
    flat.positions = pos
    dev, host = build_both(flat)
    assert dev[2]["fallbacks"] == 0 and dev[2]["device_trees"] == 1
    assert_same(dev, host, "plain zero")


def test_baseline_cfg2_tree_built_on_device_matches_reference():
    """1,000,000 triangles: the tree is the reference's; timing printed for DESIGN.md."""
    flat = ysc.plane_scene()
    t0 = time.time()
    dev, host = build_both(flat, min_prims=16384)
    print("cfg2 build info device:", dev[2], "host:", host[2], "wall", time.time() - t0)
    assert dev[2]["device_trees"] == 1 and dev[2]["device_prims"] == 1_000_000
    assert_same(dev, host, "cfg2")
    assert len(dev[0].nodes) == 503_478
    if P.have_ref():
        ref = ry.RefBvh(ry.RefScene.from_flat(flat)).flat()
        assert dev[0].nodes.tobytes() == ref.nodes.tobytes()
        assert dev[0].primitives.tobytes() == ref.primitives.tobytes()
    assert dev[2]["build_ms"] < host[2]["build_ms"]


def test_hits_and_images_do_not_depend_on_the_builder():
    flat = P.SCENES["instances"]()
    res = []
    for mode in ["device", "host"]:
        ctx = yt.Context(0)
        ctx.upload_scene(flat)
        ctx.set_bvh_builder(mode, 5)
        ctx.make_trace_bvh(flat)
        ctx.make_trace_lights(flat)
        rays = P.random_rays(flat, 50000, seed=2)
        p = yt.trace_params(sampler="path", resolution=64, samples=3, batch=3)
        res.append((ctx.intersect_batch(rays), P.gpu_render(ctx, flat, p)))
        ctx.close()
    assert res[0][0].tobytes() == res[1][0].tobytes()
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert res[0][1][k].tobytes() == res[1][1][k].tobytes(), k


# ---- highqualitybvh: split_sah (yocto_bvh.cpp:108-164) on the device --------------------------
@pytest.mark.parametrize("name", list(P.SCENES))
def test_device_sah_tree_equals_host_and_reference(name):
    flat = P.SCENES[name]()
    dev, host = build_both(flat, min_prims=5, highquality=True)
    assert dev[2]["device_trees"] >= 1 and dev[2]["fallbacks"] == 0 and host[2]["device_trees"] == 0
    assert_same(dev, host, name + " sah")
    if P.have_ref():
        ref = ry.RefBvh(ry.RefScene.from_flat(flat), True).flat()
        assert dev[0].nodes.tobytes() == ref.nodes.tobytes()
        assert dev[0].primitives.tobytes() == ref.primitives.tobytes()
    # and it is not the split_middle tree
    mid = build_both(flat, min_prims=5)[0]
    assert mid[0].nodes.tobytes() != dev[0].nodes.tobytes()


@pytest.mark.parametrize("kind", ["triangles", "quads", "lines", "points"])
@pytest.mark.parametrize("variant", ["uniform", "clustered", "grid"])
def test_device_sah_on_random_soups(kind, variant):
    """Every branch of split_sah: empty bins and one-sided candidates (0 * inf = NaN costs that
    never win), degenerate axes (csize 0 on one axis), many equal centroids (grid: candidates
    that cannot split fall to the break-in-half rule), very unbalanced clusters."""
    flat = _soup(kind, 30011, seed=len(kind) * 5 + len(variant), clustered=variant == "clustered",
                 grid=variant == "grid")
    dev, host = build_both(flat, highquality=True)
    assert dev[2]["device_trees"] == 1 and dev[2]["fallbacks"] == 0
    assert_same(dev, host, f"sah {kind}/{variant}")
    if P.have_ref() and kind == "triangles":
        ref = ry.RefBvh(ry.RefScene.from_flat(flat), True).flat()
        assert dev[0].nodes.tobytes() == ref.nodes.tobytes()
        assert dev[0].primitives.tobytes() == ref.primitives.tobytes()


def test_device_sah_degenerate_inputs():
    dev, host = build_both(_soup("triangles", 4099, seed=3, dup=True), highquality=True)
    assert dev[2]["device_trees"] == 1
    assert_same(dev, host, "sah dup")
    flat = _soup("points", 6000, seed=4)  # all centroids on a line: two axes without extent
    pos = flat.positions.copy()
    pos[:, 1] = 0.25
    pos[:, 2] = -1.5
    flat.positions = pos
    dev, host = build_both(flat, highquality=True)
    assert dev[2]["device_trees"] == 1
    assert_same(dev, host, "sah line")


def test_baseline_cfg2_sah_tree_on_device():
    """1,000,000 triangles with highqualitybvh: the host builder's tree (itself the
    reference's, tests/test_host.py) bit for bit; renders from it equal the host-built one."""
    flat = ysc.plane_scene()
    t0 = time.time()
    dev, host = build_both(flat, min_prims=16384, highquality=True)
    print("cfg2 SAH build info device:", dev[2], "host:", host[2], "wall", time.time() - t0)
    assert dev[2]["device_trees"] == 1 and dev[2]["device_prims"] == 1_000_000 and dev[2]["fallbacks"] == 0
    assert_same(dev, host, "cfg2 sah")
    assert dev[2]["build_ms"] < host[2]["build_ms"]


@pytest.mark.skipif(not P.have_ref(), reason="oracle/_ref did not travel")
@pytest.mark.parametrize("name", ["cornellbox", "materials", "lines_points", "cornell1m"])
def test_render_on_the_device_built_sah_tree_equals_the_reference_with_highqualitybvh(name):
    """trace_params::highqualitybvh end to end: tree built on the device, `path` render =
    the reference's render on ITS split_sah tree, whole trace_state byte for byte (the two
    trees order equidistant hits differently from the split_middle ones, so this is a
    parity case of its own)."""
    flat = P.scene_cornell_1m() if name == "cornell1m" else P.SCENES[name]()
    p = yt.trace_params(sampler="path", resolution=192, samples=4, batch=2)
    ctx = yt.Context(0)
    ctx.upload_scene(flat)
    ctx.set_bvh_builder("device", 5)
    ctx.make_trace_bvh(flat, True)
    ctx.make_trace_lights(flat)
    assert ctx.bvh_build_info()["device_trees"] >= 1
    gpu = P.gpu_render(ctx, flat, p)
    ctx.close()
    ref = P.RefBundle(flat, highquality=True).render(p)
    P.assert_identical(gpu, ref, name + " highqualitybvh")


# ---- round 3: the instance tree on the device, the small shapes on a host thread pool ---------------
def _many_instances(grid, rotate=True, seed=5):
    """grid x grid instances of a small sphere with random rigid + scaled frames (overlapping boxes)."""
    flat = ysc.instanced_scene(grid=grid, sphere_steps=(8, 4))
    if rotate:
        rng = np.random.default_rng(seed)
        n = len(flat.instances)
        q = rng.normal(size=(n, 3, 3))
        q, _ = np.linalg.qr(q)
        s = (0.5 + rng.random((n, 1, 1))).astype(f32)
        fr = flat.instances["frame"].copy()
        fr[:, :9] = (q * s).astype(f32).reshape(n, 9)
        flat.instances["frame"] = fr
    return flat


@pytest.mark.parametrize("highquality", [False, True])
def test_instance_tree_built_on_the_device_equals_the_host_tree(highquality):
    """40,000 instances (>= the device threshold): make_bvh over the instances' world bounds runs on
    the GPU (kind 0 of yt_gpubuild.hip) — the same nodes, the same `primitives`, the same baked
    pairs / quads as the host builder's, split_middle and split_sah; hits equal too."""
    flat = _many_instances(200)
    dev, host = build_both(flat, min_prims=16384, highquality=highquality)
    assert dev[2]["device_tlas"] == 1 and host[2]["device_tlas"] == 0
    assert_same(dev, host, "instance tree")


def test_device_instance_tree_renders_refits_and_downloads():
    flat = _many_instances(160)
    ctx = yt.Context(0)
    ctx.upload_scene(flat)
    ctx.make_trace_bvh(flat)
    assert ctx.bvh_build_info()["device_tlas"] == 1
    rays = P.random_rays(flat, 50000)
    got = ctx.intersect_batch(rays)
    hctx = yt.Context(0)
    hctx.upload_scene(flat)
    hctx.set_bvh_builder("host")
    hctx.make_trace_bvh(flat)
    want = hctx.intersect_batch(rays)
    assert P.hits_equal(want, got)
    # moving instances: the device-built instance tree is refitted on the device (kind 0 of k_refit) and stays there;
    # 3,658 frames travel as two compact arrays + one scatter kernel
    ids = np.arange(0, len(flat.instances), 7, dtype=np.int32)
    frames = flat.instances["frame"][ids].copy()
    frames[:, 10] += f32(0.05)
    for c in (ctx, hctx):
        c.update_instance_frames(ids, frames)
        c.update_bvh(updated_instances=ids)
    assert ctx.bvh_build_info()["device_tlas"] == 1 and hctx.bvh_build_info()["device_tlas"] == 0
    a, b = ctx.download_bvh(), hctx.download_bvh()
    assert a.nodes.tobytes() == b.nodes.tobytes() and a.primitives.tobytes() == b.primitives.tobytes()
    assert P.hits_equal(hctx.intersect_batch(rays), ctx.intersect_batch(rays))
    # a second edit: the tree is still on the device, an instance named twice keeps its last frame (no scatter then)
    ids2 = np.concatenate([ids[:100], ids[:3]]).astype(np.int32)
    frames2 = flat.instances["frame"][ids2].copy()
    frames2[:, 9] -= f32(0.02)
    frames2[100:, 9] -= f32(0.03)
    for c in (ctx, hctx):
        c.update_instance_frames(ids2, frames2)
        c.update_bvh(updated_instances=ids2)
    assert ctx.bvh_build_info()["device_tlas"] == 1
    a, b = ctx.download_bvh(), hctx.download_bvh()
    assert a.nodes.tobytes() == b.nodes.tobytes() and a.primitives.tobytes() == b.primitives.tobytes()
    assert P.hits_equal(hctx.intersect_batch(rays), ctx.intersect_batch(rays))
    ctx.close()
    hctx.close()


def test_small_shapes_are_built_by_a_host_thread_pool(monkeypatch):
    """1,500 shapes of 40-400 triangles each (all below the device threshold): the pool of host threads
    (the reference's parallel_for over the shapes, yocto_bvh.cpp:369-378) produces the flat layout of the
    one-thread build, byte for byte, and says how many threads it used."""
    rng = np.random.default_rng(11)
    flat = yt.FlatScene()
    flat.add_camera(ysc.lookat_frame((0, 3, 8), (0, 0, 0)))
    m = flat.add_material("matte", color=(0.6, 0.6, 0.6))
    for k in range(1500):
        n = int(rng.integers(40, 400))
        p = (rng.random((3 * n, 3)) + rng.integers(-5, 5, 3)).astype(f32)
        sh = flat.add_shape(p, triangles=np.arange(3 * n, dtype=np.int32).reshape(n, 3))
        flat.add_instance(sh, m)
    flat.add_environment((1, 1, 1))
    out = []
    for threads in ["0", "8"]:
        monkeypatch.setenv("YTHIP_BUILD_THREADS", threads)
        ctx = yt.Context(0)
        ctx.upload_scene(flat)
        ctx.make_trace_bvh(flat)
        out.append((ctx.download_bvh(), ctx.download_baked_bvh(), ctx.bvh_build_info()))
        ctx.close()
    assert out[0][2]["host_threads"] == 0 and out[1][2]["host_threads"] == 8
    assert out[0][2]["host_trees"] == out[1][2]["host_trees"] == 1500
    assert_same(out[1], out[0], "thread pool")


@pytest.mark.parametrize("name", list(P.SCENES))
def test_batched_device_bake_of_host_trees_equals_the_host_bake(name, monkeypatch):
    """Host-resident trees are baked into pair / quad / leaf records by ONE set of device launches
    (ytgpu::bake_host_trees); YTHIP_HOST_BAKE=1 assembles the same records on the host, tree by tree:
    byte-identical arrays, on every test scene (all primitive kinds, instances, empty shapes)."""
    flat = P.SCENES[name]()
    out = []
    for env in ["0", "1"]:
        monkeypatch.setenv("YTHIP_HOST_BAKE", env)
        ctx = yt.Context(0)
        ctx.upload_scene(flat)
        ctx.set_bvh_builder("host")
        ctx.make_trace_bvh(flat)
        out.append(ctx.download_baked_bvh())
        ctx.close()
    for a, b, what in zip(out[0], out[1], ["pairs", "leaf data", "quads"]):
        assert a.tobytes() == b.tobytes(), what
