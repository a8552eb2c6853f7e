"""Round-2 additions, on the GPU through the C ABI: bounces <= 0; cancellation inside a batch; in-place
material / environment edits; the one-process multi-device API (ythip_multi) sharded over
two ranks of one GPU; the deep-tree guard."""
import ctypes as C
import os
import threading
import time

import numpy as np
import pytest

import parity as P
import ythip as yt

pytestmark = pytest.mark.gpu

STATE_KEYS = ["image", "albedo", "normal", "hits", "rngs"]


def _same_state(a, b):
    return all(a[k].tobytes() == b[k].tobytes() for k in STATE_KEYS)


# ---------------------------------------------------------------------------
# bounces <= 0: the reference's bounce loop never runs
# ---------------------------------------------------------------------------
@pytest.mark.skipif(not P.have_ref(), reason="compiled reference did not travel")
@pytest.mark.parametrize("sampler", ["path", "pathdirect", "pathmis", "pathtest", "naive", "furnace", "eyelight",
                                     "falsecolor"])
@pytest.mark.parametrize("scene", ["cornellbox", "plane"])
def test_zero_bounces_match_the_reference(sampler, scene):
    """bounces = 0: radiance 0, hit false, only the camera-ray rng draws (eyelight runs
    max(bounces, 4) bounces and falsecolor none at all: unaffected) — image, guides, hits
    and rng streams as the reference leaves them (yocto_trace.cpp:466, 1045, 1260)."""
    flat = P.SCENES[scene]()
    for bounces, envhidden in [(0, False), (-3, True)]:
        params = yt.trace_params(sampler=sampler, resolution=64, samples=3, batch=2, bounces=bounces,
                                 envhidden=envhidden)
        ctx = P.gpu_context(flat)
        gpu = P.gpu_render(ctx, flat, params)
        ctx.close()
        ref = P.RefBundle(flat).render(params)
        if sampler == "falsecolor":  # srgb_to_rgb's powf: device libm vs glibc in the last ulp
            for k in ["hits", "rngs", "albedo", "normal"]:
                assert gpu[k].tobytes() == ref[k].tobytes(), (sampler, k)
            assert np.allclose(gpu["image"], ref["image"], rtol=1e-4, atol=1e-6)
        else:
            assert _same_state(gpu, ref), (sampler, scene, bounces)


# ---------------------------------------------------------------------------
# cancellation inside a batch (VERDICT r1 #4a)
# ---------------------------------------------------------------------------
def test_cancel_inside_a_batch():
    """A batch that would run for seconds (the instanced scene, 1920x1080 x 2048 spp) is
    cancelled through the caller's stop flag: the call returns YTHIP_ERR_CANCELLED within
    50 ms of the flag going up, state.samples does not advance, every pixel has taken
    between 0 and `batch` samples, and the context renders correctly afterwards."""
    import scenes as ysc
    flat = ysc.instanced_scene()
    ctx = P.gpu_context(flat)
    params = yt.trace_params(sampler="path", resolution=1920, samples=1 << 20, batch=2048)
    ctx.make_trace_state(flat, params)
    stop = np.zeros(1, np.int32)
    t_raise = [0.0]

    def raiser():
        time.sleep(0.3)
        t_raise[0] = time.perf_counter()
        stop[0] = 1

    th = threading.Thread(target=raiser)
    th.start()
    with pytest.raises(yt.YthipError) as e:
        ctx.trace_samples(params, stop=stop)
    t_back = time.perf_counter()
    th.join()
    assert "cancel" in str(e.value).lower()
    assert t_raise[0] > 0 and (t_back - t_raise[0]) < 0.05, f"returned {1e3 * (t_back - t_raise[0]):.1f} ms after the flag"
    st = ctx.download_state()
    assert st["samples"] == 0
    assert 0 <= st["hits"].min() and st["hits"].max() <= params.batch and st["hits"].max() > 0
    assert np.isfinite(st["image"]).all()
    # the next batch has another number (nothing has to lower the flag) and renders as a fresh context does
    p2 = yt.trace_params(sampler="path", resolution=160, samples=2, batch=2)
    a = P.gpu_render(ctx, flat, p2)
    ctx.close()
    ctx2 = P.gpu_context(flat)
    b = P.gpu_render(ctx2, flat, p2)
    ctx2.close()
    assert _same_state(a, b)


def test_cancel_of_an_enqueued_batch():
    import scenes as ysc
    flat = ysc.instanced_scene()
    ctx = P.gpu_context(flat)
    params = yt.trace_params(sampler="path", resolution=1920, samples=1 << 20, batch=2048)
    warm = yt.trace_params(sampler="path", resolution=1920, samples=1 << 20, batch=1)
    ctx.make_trace_state(flat, warm)
    ctx.trace_samples(warm)  # (first launch: code object load, scratch allocation)
    ctx.trace_samples_async(params)
    time.sleep(0.2)
    assert ctx.lib.ythip_poll(ctx.h) == 0, "the long batch should still be running"
    t0 = time.perf_counter()
    ctx.cancel()
    ctx.sync()
    # (seconds of work were left; the flag is written by the command processor, whose queue
    # scheduling adds up to ~0.1 s when nothing else pokes the device)
    assert time.perf_counter() - t0 < 0.5
    assert ctx.lib.ythip_poll(ctx.h) == 1
    ctx.close()


# ---------------------------------------------------------------------------
# in-place edits of the small pools
# ---------------------------------------------------------------------------
def test_update_materials_and_environments_equal_a_fresh_upload():
    flat = P.SCENES["materials"]()
    params = yt.trace_params(sampler="path", resolution=128, samples=4, batch=4)
    ctx = P.gpu_context(flat)
    P.gpu_render(ctx, flat, params)
    edited = P.SCENES["materials"]()
    edited.materials["color"][:, 0] *= 0.5
    edited.materials["roughness"] = np.minimum(edited.materials["roughness"] + 0.1, 1.0)
    edited.materials["type"][0] = edited.materials["type"][1]
    edited.environments["emission"] *= 0.7
    ctx.update_materials(edited.materials)
    ctx.update_environments(edited.environments)
    a = P.gpu_render(ctx, edited, params)
    ctx.close()
    ctx2 = P.gpu_context(edited)
    b = P.gpu_render(ctx2, edited, params)
    ctx2.close()
    assert _same_state(a, b)
    # all-matte scene whose material turns glossy: the kernel specialisation must follow
    plane = P.SCENES["plane"]()
    p2 = yt.trace_params(sampler="path", resolution=128, samples=4, batch=4)
    ctx = P.gpu_context(plane)
    P.gpu_render(ctx, plane, p2)
    glossy = P.SCENES["plane"]()
    glossy.materials["type"][:] = 2  # glossy
    glossy.materials["roughness"][:] = 0.2
    ctx.update_materials(glossy.materials)
    a = P.gpu_render(ctx, glossy, p2)
    ctx.close()
    ctx2 = P.gpu_context(glossy)
    b = P.gpu_render(ctx2, glossy, p2)
    ctx2.close()
    assert _same_state(a, b)
    with pytest.raises(yt.YthipError):
        c = P.gpu_context(plane)
        try:
            c.update_materials(np.concatenate([plane.materials, plane.materials]))
        finally:
            c.close()


# ---------------------------------------------------------------------------
# one process, several ranks: ythip_multi
# ---------------------------------------------------------------------------
def _multi_render(flat, params, devices, full_state_init=None):
    lib = yt.load_library()
    ids = (C.c_int * len(devices))(*devices)
    m = C.c_void_p()
    assert lib.ythip_create_multi(ids, len(devices), C.byref(m)) == 0, lib.ythip_multi_last_error(None)
    try:
        cs = flat.c_struct()
        for r in range(lib.ythip_multi_size(m)):
            ctx = C.c_void_p(lib.ythip_multi_ctx(m, r))
            assert lib.ythip_upload_scene(ctx, C.byref(cs)) == 0
            assert lib.ythip_build_bvh(ctx, C.byref(cs), 0) == 0
            assert lib.ythip_build_lights(ctx, C.byref(cs)) == 0
        w, h = yt.state_size(flat.cameras[0], params.resolution)
        assert lib.ythip_multi_state_create(m, w, h) == 0, lib.ythip_multi_last_error(m)
        rngs = yt.make_rngs(params.seed, w * h)
        assert lib.ythip_multi_state_upload(m, None, None, None, None, rngs.ctypes.data, 0) == 0
        n = 0
        while n < params.samples:
            assert lib.ythip_multi_trace_samples(m, C.byref(params), None) == 0, lib.ythip_multi_last_error(m)
            n += params.batch
        st = dict(image=np.zeros((w * h, 4), np.float32), albedo=np.zeros((w * h, 3), np.float32),
                  normal=np.zeros((w * h, 3), np.float32), hits=np.zeros(w * h, np.int32),
                  rngs=np.zeros((w * h, 2), np.uint64))
        s = C.c_int()
        assert lib.ythip_multi_state_download(m, st["image"].ctypes.data, st["albedo"].ctypes.data,
                                              st["normal"].ctypes.data, st["hits"].ctypes.data,
                                              st["rngs"].ctypes.data, C.byref(s)) == 0
        img = np.zeros((w * h, 4), np.float32)
        assert lib.ythip_multi_get_image(m, img.ctypes.data) == 0, lib.ythip_multi_last_error(m)
        mode = C.create_string_buffer(64)
        ranks = C.c_int()
        lib.ythip_multi_gather_info(m, mode, 64, C.byref(ranks))
        return st, img, s.value, mode.value.decode(), ranks.value
    finally:
        lib.ythip_destroy_multi(m)


@pytest.mark.parametrize("nranks", [1, 2, 3])
def test_multi_device_api_equals_one_context(nranks):
    """ythip_create_multi({0, 0, ...}): the frame's tile columns dealt over `nranks` ranks
    (here all on device 0), per-rank contexts, concurrent asynchronous launches, full-frame
    upload / download through the slices and the gathered framebuffer — equal, byte for
    byte, to one context rendering the whole frame (odd frame width: ragged last column)."""
    flat = P.SCENES["plane"]()
    params = yt.trace_params(sampler="path", resolution=300, samples=4, batch=2)
    ctx = P.gpu_context(flat)
    one = P.gpu_render(ctx, flat, params)
    ctx.close()
    st, img, samples, mode, ranks = _multi_render(flat, params, [0] * nranks)
    assert samples == 4
    for k in STATE_KEYS:
        assert st[k].tobytes() == one[k].tobytes(), k
    assert img.tobytes() == one["image"].tobytes()
    assert mode == ("none (one rank)" if nranks == 1 else "device copies") and ranks == 0


@pytest.mark.parametrize("nranks", [2, 4])
def test_multi_device_gather_equals_the_process_per_gpu_gather(nranks):
    """The two multi-GPU launch models assemble the same frame: `ythip_multi` (one process, its
    own gather — here the device-copy branch: RCCL refuses a communicator that names a device twice) against what
    `bench.py --gpus N` does — one plain context per rank on its striped slice of the frame, the slices put in place
    by sharding.FrameGather's permutation (the one the gloo tests of tests/test_sharding.py exercise)."""
    from sharding import shard_frame
    flat = P.SCENES["materials"]()
    params = yt.trace_params(sampler="path", resolution=272, samples=4, batch=4)  # 17 tile columns: uneven stripes
    st, img, samples, mode, _ = _multi_render(flat, params, [0] * nranks)
    assert mode == "device copies"
    w, h = yt.state_size(flat.cameras[0], params.resolution)
    rngs = yt.make_rngs(params.seed, w * h)
    frame = np.zeros((w * h, 4), np.float32)
    covered = np.zeros(w * h, bool)
    for r in range(nranks):
        shard = shard_frame(w, h, nranks, r, "columns")
        ctx = P.gpu_context(flat)
        ctx.make_trace_state(flat, params, rows=shard.rows, cols=shard.cols, rngs=rngs)
        ctx.trace_samples(params)
        frame[shard.pixels] = ctx.download_state()["image"].reshape(-1, 4)
        covered[shard.pixels] = True
        ctx.close()
    assert covered.all()
    assert frame.tobytes() == img.tobytes() == st["image"].tobytes()


def test_rccl_send_recv_branch_executes_on_one_device(monkeypatch):
    """The framebuffer gather's RCCL branch (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd,
    yt_multi.hip) on the hardware there is: YTHIP_GATHER=rccl-self builds a one-rank communicator
    on device 0 and sends rank 0's slice to itself inside the group.  The gathered frame equals a
    single context's, and ythip_multi_gather_info reports the path that ran + RCCL's rank count."""
    monkeypatch.setenv("YTHIP_GATHER", "rccl-self")
    flat = P.SCENES["plane"]()
    params = yt.trace_params(sampler="path", resolution=300, samples=4, batch=2)
    ctx = P.gpu_context(flat)
    one = P.gpu_render(ctx, flat, params)
    ctx.close()
    st, img, samples, mode, ranks = _multi_render(flat, params, [0])
    assert samples == 4
    assert img.tobytes() == one["image"].tobytes()
    assert mode == "rccl send/recv (incl. rank 0 to itself)" and ranks == 1


def test_multi_device_api_errors():
    lib = yt.load_library()
    m = C.c_void_p()
    assert lib.ythip_create_multi(None, 2, C.byref(m)) != 0
    ids = (C.c_int * 2)(0, 9999)
    assert lib.ythip_create_multi(ids, 2, C.byref(m)) != 0
    assert b"rank 1" in lib.ythip_multi_last_error(None)
    ids = (C.c_int * 2)(0, 0)
    assert lib.ythip_create_multi(ids, 2, C.byref(m)) == 0
    assert lib.ythip_multi_state_create(m, 0, 16) != 0
    assert lib.ythip_multi_state_create(m, 16, 16) == 0  # one tile column for two ranks: rank 1 sits out
    assert lib.ythip_multi_get_image(m, None) != 0
    lib.ythip_destroy_multi(m)


# ---- deep trees: the shared 128-entry stack ------------------------------------------------------
def _chain_scene_and_tree(n):
    """n well-separated triangles along x and a hand-made DEGENERATE tree over them: a chain in
    which every internal node has one leaf child and one internal child (depth n) — what a
    pathological builder could hand to upload_bvh."""
    flat = yt.FlatScene()
    flat.add_camera(yt.IDENTITY_FRAME)
    x = np.arange(n, dtype="f4")[:, None] * 2.0
    p0 = np.concatenate([x, np.zeros((n, 2), "f4")], 1)
    positions = np.concatenate([p0, p0 + [1, 0, 0], p0 + [0, 1, 0]]).astype("f4")
    tris = np.stack([np.arange(n), np.arange(n) + n, np.arange(n) + 2 * n], 1).astype("i4")
    sh = flat.add_shape(positions, triangles=tris)
    m = flat.add_material(color=(0.5, 0.5, 0.5))
    flat.add_instance(sh, m)
    flat.add_environment((1, 1, 1))
    lo = np.minimum(np.minimum(positions[tris[:, 0]], positions[tris[:, 1]]), positions[tris[:, 2]])
    hi = np.maximum(np.maximum(positions[tris[:, 0]], positions[tris[:, 1]]), positions[tris[:, 2]])
    nodes = np.zeros(2 * n - 1, yt.node_dt)
    # node 2k: internal (k < n - 1) covering triangles k..n-1; node 2k + 1: leaf of triangle k;
    # the last internal node's second child is the leaf of triangle n - 1
    for k in range(n - 1):
        nd = nodes[2 * k]
        nd["bbox_min"], nd["bbox_max"] = lo[k:].min(0), hi[k:].max(0)
        nd["start"], nd["num"], nd["axis"], nd["internal"] = 2 * k + 1, 2, 0, 1
        lf = nodes[2 * k + 1]
        lf["bbox_min"], lf["bbox_max"], lf["start"], lf["num"] = lo[k], hi[k], k, 1
    last = nodes[2 * n - 2]
    last["bbox_min"], last["bbox_max"], last["start"], last["num"] = lo[n - 1], hi[n - 1], n - 1, 1
    tl = np.zeros(1, yt.node_dt)
    tl["bbox_min"], tl["bbox_max"], tl["start"], tl["num"] = lo.min(0), hi.max(0), 0, 1
    bvh = yt.FlatBvh([0, 2 * n - 1, 2 * n], [0, n, n + 1], np.concatenate([nodes, tl]),
                     np.concatenate([np.arange(n, dtype="i4"), np.zeros(1, "i4")]))
    return flat, bvh


def test_trees_too_deep_for_the_wide_walk_are_walked_binary_and_deeper_ones_refused():
    """One 128-entry stack serves the instance tree and the shape tree.  A chain of depth 100 passes
    the reference-shaped bound (100 + 1 + 5 <= 128) but not the wide walk's (3 pending siblings per
    two levels: ADVICE r2) — it is walked binary, with the hit records of the builder's balanced
    tree over the same triangles.  Depth 130 would overflow the reference's own stack: refused."""
    flat, chain = _chain_scene_and_tree(100)
    rays = P.random_rays(flat, 20000)
    ctx = P.gpu_context(flat)
    want = ctx.intersect_batch(rays)
    ctx.upload_bvh(chain)
    for mode in ["auto", "wide", "binary"]:
        ctx.set_traversal(mode)
        got = ctx.intersect_batch(rays)
        assert P.hits_equal(want, got), mode
    ctx.close()
    assert int(want["hit"].sum()) > 100
    flat, chain = _chain_scene_and_tree(130)
    ctx = P.gpu_context(flat)
    with pytest.raises(yt.YthipError) as e:
        ctx.upload_bvh(chain)
    assert "too deep" in str(e.value)
    ctx.close()


# ---- random scenes x random parameters vs the live reference -------------------------------------
@pytest.mark.skipif(not P.have_ref(), reason="oracle/_ref did not travel")
def test_fuzz_random_scenes_against_the_live_reference():
    """tools/fuzz_parity.py: 80 random scenes (all primitive kinds, material types, texture
    slots, environments, non-rigid frames, cameras) x random trace_params (all samplers,
    bounces 0-8, clamp, nocaustics, envhidden, tentfilter, batches, split_middle / split_sah):
    whole trace_state byte for byte.  (profiles/r02_fuzz.txt: 4,000 cases, 0 failures.)"""
    import subprocess, sys
    tool = os.path.join(P.ROOT, "tools", "fuzz_parity.py")
    r = subprocess.run([sys.executable, tool, "5000", "80"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "80 cases, 0 failures" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


# ---- longest-tile-first launch order (yt_order.hip) ------------------------------------------------
@pytest.mark.parametrize("scene,sampler", [("cornellbox", "path"), ("materials", "path"), ("lines_points", "pathdirect"),
                                           ("instances", "naive")])
def test_launch_order_by_measured_tile_cost_changes_nothing(scene, sampler, monkeypatch):
    """From the second batch on k_trace hands out the tiles most expensive first (the costs
    every workgroup recorded in the previous launch, sorted on the device).  Only the tile ->
    workgroup assignment changes: progressive renders with the order on and off, a
    trace_sample between batches and a column-sliced state all give the same bytes."""
    flat = P.SCENES[scene]()
    p = yt.trace_params(sampler=sampler, resolution=200, samples=8, batch=2)
    out = []
    for lpt in ["1", "0"]:
        monkeypatch.setenv("YTHIP_LPT", lpt)
        ctx = P.gpu_context(flat)
        ctx.make_trace_state(flat, p)
        ctx.trace_samples(p)
        ctx.trace_samples(p)
        ctx.trace_sample(p, 17, 5, 4)  # (single-pixel launches do not disturb the recorded costs)
        ctx.trace_samples(p)
        ctx.trace_samples(p)
        full = ctx.download_state()
        sl = P.gpu_render(ctx, flat, p, cols=(1, 3))
        out.append((full, sl))
        ctx.close()
    P.assert_identical(out[0][0], out[1][0], f"{scene} {sampler} full frame")
    P.assert_identical(out[0][1], out[1][1], f"{scene} {sampler} slice")
