"""update_scene_bvh on the device ("build/refit",
yocto_bvh.cpp:304-319, 398-451; k_refit in yt_gpubuild.hip), `-m gpu`.

After an edit that keeps the element lists the resident trees are refitted where
they live (device-built trees by one bottom-up kernel, the rest on the host) and
re-baked.  Bar: the refitted tree is the reference's update_scene_bvh result bit
for bit, the baked traversal arrays equal those of a from-scratch upload of the
same tree, and hits / images on the edited scene equal the reference's."""
import time

import numpy as np
import pytest

import parity as P
from parity import ry, yt, ysc
from test_gpu_build import _soup

pytestmark = pytest.mark.gpu
f32 = np.float32


def refit_both(flat, edited, shapes, insts, min_prims=5):
    """The same edit through the device refit and through the host refit."""
    out = []
    for mode in ["device", "host"]:
        ctx = yt.Context(0)
        ctx.upload_scene(flat)
        ctx.set_bvh_builder(mode, min_prims)
        ctx.make_trace_bvh(flat)
        P.apply_edit_to_gpu(ctx, edited, shapes, insts)
        ctx.update_bvh(insts, shapes)
        out.append((ctx.download_bvh(), ctx.download_baked_bvh(), ctx.bvh_build_info()))
        ctx.close()
    return out


def assert_same(dev, host, what=""):
    (db, (dp, dl, dq), _), (hb, (hp, hl, hq), _) = dev, host
    assert db.primitives.tobytes() == hb.primitives.tobytes(), what + " primitives"
    assert db.nodes.tobytes() == hb.nodes.tobytes(), what + " nodes"
    assert dp.tobytes() == hp.tobytes(), what + " baked pairs"
    assert dl.tobytes() == hl.tobytes(), what + " baked leaf data"
    assert dq.tobytes() == hq.tobytes(), what + " baked quads"


@pytest.mark.parametrize("name", list(P.SCENES))
def test_device_refit_equals_host_refit_and_reference(name):
    flat = P.SCENES[name]()
    edited, shapes, insts = P.edit_scene(flat)
    dev, host = refit_both(flat, edited, shapes, insts)
    assert dev[2]["device_trees"] >= 1 and host[2]["device_trees"] == 0
    assert_same(dev, host, name)
    assert dev[0].nodes.tobytes() == yt.host_update_bvh(flat, edited, shapes).nodes.tobytes()
    if P.have_ref():
        rs = ry.RefScene.from_flat(flat)
        rb = ry.RefBvh(rs)
        P.apply_edit_to_ref(rs, edited, shapes, insts)
        rb.update(rs, insts, shapes)
        assert dev[0].nodes.tobytes() == rb.flat().nodes.tobytes()


@pytest.mark.parametrize("kind", ["triangles", "quads", "lines", "points"])
def test_device_refit_on_random_soups(kind):
    flat = _soup(kind, 30011, seed=len(kind))
    edited, shapes, insts = P.edit_scene(flat, seed=5, shapes=[0], amount=0.3)
    dev, host = refit_both(flat, edited, shapes, insts, min_prims=8)
    assert dev[2]["device_trees"] == 1
    assert_same(dev, host, kind)


def test_refitted_scene_traces_like_a_fresh_upload_of_the_same_tree_and_like_the_reference():
    flat = P.SCENES["instances"]()
    edited, shapes, insts = P.edit_scene(flat, seed=3)
    ctx = yt.Context(0)
    ctx.upload_scene(flat)
    ctx.set_bvh_builder("device", 5)
    ctx.make_trace_bvh(flat)
    P.apply_edit_to_gpu(ctx, edited, shapes, insts)
    ctx.update_bvh(insts, shapes)
    ctx.make_trace_lights(edited)
    rays = P.random_rays(edited, 50000, seed=4)
    p = yt.trace_params(sampler="path", resolution=64, samples=3, batch=3)
    hits, img = ctx.intersect_batch(rays), P.gpu_render(ctx, edited, p)
    tree = ctx.download_bvh()
    ctx.close()
    # the edited scene uploaded from scratch with that (refitted, not rebuilt) tree
    ctx = yt.Context(0)
    ctx.upload_scene(edited)
    ctx.upload_bvh(tree)
    ctx.make_trace_lights(edited)
    hits2, img2 = ctx.intersect_batch(rays), P.gpu_render(ctx, edited, p)
    ctx.close()
    assert hits.tobytes() == hits2.tobytes()
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert img[k].tobytes() == img2[k].tobytes(), k
    if P.have_ref():
        rs = ry.RefScene.from_flat(flat)
        rb = ry.RefBvh(rs)
        P.apply_edit_to_ref(rs, edited, shapes, insts)
        rb.update(rs, insts, shapes)
        assert P.hits_equal(hits, ry.intersect_batch(rb, rs, rays))
        st = ry.RefState(rs, p)
        lights = ry.RefLights(rs)
        while st.samples < p.samples:
            ry.trace_samples(st, rs, rb, lights, p)
        ref = st.get()
        assert img["rngs"].tobytes() == ref["rngs"].tobytes()
        assert img["hits"].tobytes() == ref["hits"].tobytes()
        P.assert_identical(img, ref, "render of the refitted scene")


def test_update_keeps_working_over_several_edits_and_rejects_bad_arguments():
    flat = P.SCENES["lines_points"]()
    ctx = yt.Context(0)
    ctx.upload_scene(flat)
    ctx.set_bvh_builder("device", 5)
    ctx.make_trace_bvh(flat)
    cur = flat
    for step in range(3):
        cur, shapes, insts = P.edit_scene(cur, seed=20 + step, shapes=list(range(len(flat.shapes))))
        P.apply_edit_to_gpu(ctx, cur, shapes, insts)
        ctx.update_bvh(insts, shapes)
        assert ctx.download_bvh().nodes.tobytes() == yt.host_update_bvh(flat, cur, shapes).nodes.tobytes(), step
    with pytest.raises(yt.YthipError):
        ctx.update_bvh([], [len(flat.shapes)])
    with pytest.raises(yt.YthipError):
        ctx.update_shape_vertices(0, np.zeros((1, 3), f32))
    with pytest.raises(yt.YthipError):
        ctx.update_instance_frames([len(flat.instances)], np.zeros((1, 12), f32))
    ctx.close()


def test_baseline_cfg2_refit_on_device():
    """1,000,000 triangles displaced: the refitted tree is the host sweep's (and the
    reference's), timing printed for DESIGN.md."""
    flat = ysc.plane_scene()
    edited, shapes, insts = P.edit_scene(flat, seed=2, shapes=[0], amount=0.001)
    t0 = time.time()
    dev, host = refit_both(flat, edited, shapes, insts, min_prims=16384)
    print("cfg2 refit info device:", dev[2], "host:", host[2], "wall", time.time() - t0)
    assert dev[2]["device_trees"] == 1 and dev[2]["device_prims"] == 1_000_000
    assert_same(dev, host, "cfg2")
    if P.have_ref():
        rs = ry.RefScene.from_flat(flat)
        rb = ry.RefBvh(rs)
        P.apply_edit_to_ref(rs, edited, shapes, insts)
        dt = rb.update(rs, insts, shapes)
        print("reference update_scene_bvh: %.1f ms" % (dt * 1e3))
        assert dev[0].nodes.tobytes() == rb.flat().nodes.tobytes()
