"""update_scene_bvh / refit_bvh (yocto_bvh.cpp:304-319,
398-451) — host half, runs without a GPU.

An edit that moves vertices and instances but keeps the element lists must leave
the tree's topology and `primitives` alone and recompute every box bottom-up,
bit for bit what the reference's sweep produces (the merge order decides the sign
of a zero face, so "the same set" is not enough).  Pinned against the live
reference's update_scene_bvh when it is here, and against the definition of a
refitted tree always."""
import numpy as np
import pytest

import parity as P
from parity import ry, yt

f32 = np.float32


def _prim_bounds(flat, s):
    """Per-primitive boxes of shape s in numpy (yocto_geometry.h:475-498)."""
    a = flat.shape_arrays(s)
    pos = a["positions"]
    rad = a["radius"].reshape(-1) if a["radius"] is not None else None
    if a["points"] is not None:
        v = a["points"].reshape(-1)
        lo, hi = pos[v] - rad[v, None], pos[v] + rad[v, None]
        return np.minimum(lo, hi), np.maximum(lo, hi)
    if a["lines"] is not None:
        l = a["lines"]
        return (np.minimum(pos[l[:, 0]] - rad[l[:, 0], None], pos[l[:, 1]] - rad[l[:, 1], None]),
                np.maximum(pos[l[:, 0]] + rad[l[:, 0], None], pos[l[:, 1]] + rad[l[:, 1], None]))
    el = a["triangles"] if a["triangles"] is not None else a["quads"]
    return pos[el].min(axis=1), pos[el].max(axis=1)


def check_is_refit_of(bvh, before, flat, shapes):
    """Topology untouched; leaves bound their primitives, internals their children."""
    assert np.array_equal(bvh.node_offset, before.node_offset)
    assert bvh.primitives.tobytes() == before.primitives.tobytes()
    for f in ["start", "num", "axis", "internal"]:
        assert np.array_equal(bvh.nodes[f], before.nodes[f]), f
    for s in shapes:
        n0, n1 = int(bvh.node_offset[s]), int(bvh.node_offset[s + 1])
        nodes = bvh.nodes[n0:n1]
        prims = bvh.primitives[int(bvh.prim_offset[s]):int(bvh.prim_offset[s + 1])]
        lo, hi = _prim_bounds(flat, s)
        for k in range(len(nodes) - 1, -1, -1):
            nd = nodes[k]
            if nd["internal"]:
                c = nodes[int(nd["start"]):int(nd["start"]) + 2]
                emin, emax = c["bbox_min"].min(axis=0), c["bbox_max"].max(axis=0)
            else:
                ids = prims[int(nd["start"]):int(nd["start"]) + int(nd["num"])]
                if len(ids) == 0:
                    continue
                emin, emax = lo[ids].min(axis=0), hi[ids].max(axis=0)
            assert np.array_equal(nd["bbox_min"], emin) and np.array_equal(nd["bbox_max"], emax), (s, k)


@pytest.mark.parametrize("name", list(P.SCENES))
def test_host_refit_is_a_refit_and_equals_reference(name):
    flat = P.SCENES[name]()
    edited, shapes, insts = P.edit_scene(flat)
    before = yt.host_make_bvh(flat)
    after = yt.host_update_bvh(flat, edited, shapes)
    assert after.nodes.tobytes() != before.nodes.tobytes()  # the edit did move boxes
    check_is_refit_of(after, before, edited, shapes)
    if not P.have_ref():
        pytest.skip("oracle/_ref not built: reference comparison skipped")
    rs = ry.RefScene.from_flat(flat)
    rb = ry.RefBvh(rs)
    assert rb.flat().nodes.tobytes() == before.nodes.tobytes()
    P.apply_edit_to_ref(rs, edited, shapes, insts)
    rb.update(rs, insts, shapes)
    ref = rb.flat()
    assert ref.primitives.tobytes() == after.primitives.tobytes()
    assert ref.nodes.tobytes() == after.nodes.tobytes(), name


def test_host_refit_with_nothing_listed_only_moves_the_instance_tree():
    flat = P.SCENES["instances"]()
    edited, _, insts = P.edit_scene(flat, shapes=[])
    before = yt.host_make_bvh(flat)
    after = yt.host_update_bvh(flat, edited, [])
    t = len(flat.shapes)
    cut = int(before.node_offset[t])
    assert after.nodes[:cut].tobytes() == before.nodes[:cut].tobytes()
    assert after.nodes[cut:].tobytes() != before.nodes[cut:].tobytes()
    if P.have_ref():
        rs = ry.RefScene.from_flat(flat)
        rb = ry.RefBvh(rs)
        P.apply_edit_to_ref(rs, edited, [], insts)
        rb.update(rs, insts, [])
        assert rb.flat().nodes.tobytes() == after.nodes.tobytes()


def test_host_refit_rejects_bad_arguments():
    flat = P.SCENES["cornellbox"]()
    with pytest.raises(yt.YthipError):
        yt.host_update_bvh(flat, flat, [len(flat.shapes)])
    other = P.SCENES["plane"]()
    with pytest.raises(yt.YthipError):
        yt.host_update_bvh(flat, other, [0])


def test_refit_sign_of_zero_follows_the_merge_order():
    """Two triangles whose boxes tie at a zero face with opposite signs: the box of
    their leaf takes the sign the reference's left-to-right merge leaves."""
    sc = yt.FlatScene()
    sc.add_camera(np.asarray(yt.IDENTITY_FRAME, "f4"))
    m = sc.add_material("matte", color=(0.5, 0.5, 0.5))
    pos = np.array([[1, 1, 1], [2, 1, 1], [1, 2, 1], [1, 1, 2], [2, 1, 2], [1, 2, 2]], f32)
    s = sc.add_shape(pos, triangles=np.array([[0, 1, 2], [3, 4, 5]], np.int32))
    sc.add_instance(s, m)
    import copy
    ed = copy.deepcopy(sc)
    ed.positions[0:3, 0] = [f32(0.0), 2, 1]   # first triangle: min.x = +0
    ed.positions[3:6, 0] = [f32(-0.0), 2, 1]  # second: min.x = -0
    after = yt.host_update_bvh(sc, ed, [0])
    root = after.nodes[0]
    assert not root["internal"] and root["num"] == 2
    # merge(merge(invalid, +0), -0): min(+0, -0) = (+0 < -0) ? +0 : -0 = -0
    assert np.signbit(root["bbox_min"][0]) and root["bbox_min"][0] == 0
    ed.positions[0:3, 0] = [f32(-0.0), 2, 1]
    ed.positions[3:6, 0] = [f32(0.0), 2, 1]
    root = yt.host_update_bvh(sc, ed, [0]).nodes[0]
    assert not np.signbit(root["bbox_min"][0]) and root["bbox_min"][0] == 0
    if P.have_ref():
        rs = ry.RefScene.from_flat(sc)
        rb = ry.RefBvh(rs)
        P.apply_edit_to_ref(rs, ed, [0], [])
        rb.update(rs, [], [0])
        assert rb.flat().nodes.tobytes() == yt.host_update_bvh(sc, ed, [0]).nodes.tobytes()
