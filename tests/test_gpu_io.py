"""Scene ingest from FILES straight into the pinned staging pools (`-m gpu`): the
shapes of a scene saved by the reference (scene.json + shapes/*.ply) are opened with ythip_ply_open, the
pools of ythip_scene_staging are sized from their counts, ythip_ply_read converts every file into the
pools at its shape's offsets, ythip_upload_scene_staged sends them — and the render equals the render of
the same scene loaded by the reference's own load_scene and uploaded the ordinary way, bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

import parity as P
from parity import ry, yt

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not P.have_ref(), reason="oracle/_ref did not travel")]


@pytest.mark.parametrize("name", ["cornellbox", "materials"])
def test_ply_files_into_the_staging_pools(name, tmp_path):
    src = ry.RefScene.from_flat(P.SCENES[name]())
    scene_file = str(tmp_path / "scene.json")
    src.save(scene_file)
    flat = ry.RefScene.load(scene_file).flat()  # what the reference's loader makes of its own files
    plys = sorted(os.listdir(tmp_path / "shapes"))
    assert len(plys) == len(flat.shapes)
    lib = yt.load_library()

    # pass 1: counts
    handles, counts = [], np.zeros(len(plys), yt.shape_dt)
    for k, f in enumerate(plys):
        h = C.c_void_p()
        assert lib.ythip_ply_open(str(tmp_path / "shapes" / f).encode(), C.byref(h), counts[k:].ctypes.data) == 0, \
            lib.ythip_io_last_error()
        handles.append(h)
    want = yt.CScene()
    want.num_cameras, want.num_instances = len(flat.cameras), len(flat.instances)
    want.num_environments, want.num_shapes = len(flat.environments), len(plys)
    want.num_textures, want.num_materials = len(flat.textures), len(flat.materials)
    for name_, _, _ in yt.FlatScene.POOLS[:9]:
        setattr(want, "num_" + name_, int(counts["num_" + name_].sum()))
    want.num_pixelsf, want.num_pixelsb = len(flat.pixelsf), len(flat.pixelsb)
    ctx = yt.Context(0)
    staged = yt.CScene()
    assert lib.ythip_scene_staging(ctx.h, C.byref(want), C.byref(staged)) == 0

    # pass 2: every file into the pools at its shape's offsets; the small pools as the loader has them
    def view(ptr, n, dt):
        return np.frombuffer((C.c_char * max(n * np.dtype(dt).itemsize, 1)).from_address(ptr), dt, n)
    cursor = {n: 0 for n, _, _ in yt.FlatScene.POOLS[:9]}
    shapes = view(staged.shapes, len(plys), yt.shape_dt)
    for k, h in enumerate(handles):
        args, rec = {}, counts[k].copy()
        for n, dt, w in yt.FlatScene.POOLS[:9]:
            cnt = int(counts[k]["num_" + n])
            rec[n + "_offset"] = cursor[n] if cnt else -1
            args[n] = getattr(staged, n) + cursor[n] * w * np.dtype(dt).itemsize if cnt else None
            cursor[n] += cnt
        shapes[k] = rec
        assert lib.ythip_ply_read(h, 1, args["positions"], args["normals"], args["texcoords"], args["colors"], args["radius"],
                                  args["points"], args["lines"], args["triangles"], args["quads"]) == 0
        lib.ythip_ply_close(h)
    view(staged.cameras, len(flat.cameras), yt.camera_dt)[:] = flat.cameras
    view(staged.instances, len(flat.instances), yt.instance_dt)[:] = flat.instances
    view(staged.environments, len(flat.environments), yt.environment_dt)[:] = flat.environments
    view(staged.materials, len(flat.materials), yt.material_dt)[:] = flat.materials
    view(staged.textures, len(flat.textures), yt.texture_dt)[:] = flat.textures
    if len(flat.pixelsf):
        view(staged.pixelsf, flat.pixelsf.size, "f4")[:] = flat.pixelsf.reshape(-1)
    if len(flat.pixelsb):
        view(staged.pixelsb, flat.pixelsb.size, "u1")[:] = flat.pixelsb.reshape(-1)
    # the pools equal the reference loader's flattened scene, byte for byte
    for n, dt, w in yt.FlatScene.POOLS[:9]:
        got = view(getattr(staged, n), cursor[n] * w, dt)
        assert got.tobytes() == np.ascontiguousarray(getattr(flat, n)).tobytes(), n
    assert shapes.tobytes() == np.ascontiguousarray(flat.shapes).tobytes()
    assert lib.ythip_upload_scene_staged(ctx.h) == 0
    ctx.make_trace_bvh(flat)
    ctx.make_trace_lights(flat)
    params = yt.trace_params(sampler="path", resolution=96, samples=4, batch=2)
    got = P.gpu_render(ctx, flat, params)
    ctx.close()
    ref = P.gpu_context(flat)
    want_state = P.gpu_render(ref, flat, params)
    ref.close()
    P.assert_identical(want_state, got, name)


@pytest.mark.parametrize("name", ["materials", "lines_points"])
def test_load_scene_files_to_hbm(name, tmp_path):
    """ythip_load_scene: scene.json + PLY + PNG / HDR files -> pinned staging pools -> HBM in one call; the render
    equals the render of the reference loader's scene uploaded the ordinary way, and the staged pools are the
    reference loader's flattened scene."""
    src = ry.RefScene.from_flat(P.SCENES[name]())
    scene_file = str(tmp_path / "scene.json")
    src.save(scene_file)
    flat = ry.RefScene.load(scene_file).flat()
    ctx = yt.Context(0)
    staged = ctx.load_scene(scene_file)
    for field in P._FIELDS:
        assert np.ascontiguousarray(getattr(staged, field)).tobytes() == np.ascontiguousarray(getattr(flat, field)).tobytes(), field
    ctx.make_trace_bvh(staged)
    ctx.make_trace_lights(staged)
    params = yt.trace_params(sampler="path", resolution=96, samples=4, batch=2)
    got = P.gpu_render(ctx, staged, params)
    ctx.close()
    ref = P.gpu_context(flat)
    want = P.gpu_render(ref, flat, params)
    ref.close()
    P.assert_identical(want, got, name)


def test_load_scene_refusal_leaves_the_context_usable(tmp_path):
    ctx = yt.Context(0)
    with pytest.raises(yt.YthipError, match="cannot open"):
        ctx.load_scene(str(tmp_path / "absent.json"))
    # a failure AFTER the pools were staged (a PNG whose header is fine and whose pixel data is not): nothing half-read is uploadable
    scene_file = str(tmp_path / "scene.json")
    ry.RefScene.from_flat(P.SCENES["materials"]()).save(scene_file)
    png = [f for f in os.listdir(tmp_path / "textures") if f.endswith(".png")][0]
    data = bytearray(open(tmp_path / "textures" / png, "rb").read())
    at = data.index(b"IDAT") + 6
    data[at:at + 8] = b"\xff" * 8
    open(tmp_path / "textures" / png, "wb").write(bytes(data))
    with pytest.raises(yt.YthipError, match="cannot raed"):
        ctx.load_scene(scene_file)
    assert ctx.lib.ythip_upload_scene_staged(ctx.h) == 0  # (what is staged now is an empty scene)
    flat = P.SCENES["cornellbox"]()
    ctx.upload_scene(flat)
    ctx.make_trace_bvh(flat)
    ctx.make_trace_lights(flat)
    P.gpu_render(ctx, flat, yt.trace_params(sampler="eyelight", resolution=32, samples=1, batch=1))
    ctx.close()


def test_staging_pools_are_reused_and_regrown(tmp_path):
    """Three loads into one context — a small scene, a larger one (pools regrown), the small one again (pools
    reused: the staged pointers are the larger load's) — each renders what a fresh context renders."""
    import ctypes as C
    files = {}
    for name in ("cornellbox", "materials"):
        os.mkdir(tmp_path / name)
        files[name] = str(tmp_path / name / "scene.json")
        ry.RefScene.from_flat(P.SCENES[name]()).save(files[name])
    params = yt.trace_params(sampler="path", resolution=64, samples=2, batch=2)
    ctx = yt.Context(0)
    pointers = []
    for name in ("cornellbox", "materials", "cornellbox"):
        staged = yt.CScene()
        assert ctx.lib.ythip_load_scene(ctx.h, files[name].encode(), 0, C.byref(staged)) == 0, ctx.lib.ythip_io_last_error()
        pointers.append(staged.positions)
        flat = yt.FlatScene.from_c(staged)
        ctx.make_trace_bvh(flat)
        ctx.make_trace_lights(flat)
        got = P.gpu_render(ctx, flat, params)
        fresh = P.gpu_context(flat)
        want = P.gpu_render(fresh, flat, params)
        fresh.close()
        P.assert_identical(want, got, name)
    assert pointers[2] == pointers[1]
    ctx.close()
