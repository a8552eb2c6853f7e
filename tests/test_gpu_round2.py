"""Round-2 additions, on the GPU through the C ABI: the pool scheduler (k_pool) gives
k_trace's trace_state bit for bit; bounces <= 0; cancellation inside a batch; in-place
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
# k_pool == k_trace
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,res,spp", [("plane", 320, 8), ("materials", 160, 6), ("lines_points", 128, 6),
                                          ("instances", 192, 6), ("cornell1m", 192, 6)])
@pytest.mark.parametrize("sampler", ["path", "pathtest"])
def test_pool_scheduler_is_bit_identical(name, res, spp, sampler):
    """Same per-path arithmetic, another schedule (yt_pool.h): dynamic ray fetch, parked
    walks, out-of-line shade passes, tiles from a global counter — and the same state.
    Few wavefronts (many tiles each), several refill / in-flight settings, progressive
    batches; cornell1m has area lights (deferred light-pdf passes), materials every lobe +
    textures + volumes (general kernel), lines_points the non-triangle leaf kinds."""
    flat = P.scene_cornell_1m(n=64) if name == "cornell1m" else P.SCENES[name]()
    params = yt.trace_params(sampler=sampler, resolution=res, samples=spp, batch=spp // 2)
    ctx = P.gpu_context(flat)
    ctx.set_traversal("wide")  # (k_pool is the wide walk; tiny test trees would otherwise stay binary)
    ctx.set_pool(0)
    base = P.gpu_render(ctx, flat, params)
    for waves, target, refill, shade_min in [(0, 128, 16, 64), (24, 192, 4, 32), (7, 64, 64, 8)]:
        ctx.set_pool(1, waves, target, refill, shade_min)
        got = P.gpu_render(ctx, flat, params)
        stats, _ = ctx.pool_stats()
        assert stats["iters"] > 0, "k_pool did not run"
        assert stats["watchdog"] == 0
        assert _same_state(base, got), (name, sampler, waves, target, refill, shade_min)
    ctx.close()


def test_pool_scheduler_row_and_column_slices():
    flat = P.SCENES["plane"]()
    params = yt.trace_params(sampler="path", resolution=256, samples=4, batch=4)
    w, h = yt.state_size(flat.cameras[0], params.resolution)
    rngs = yt.make_rngs(params.seed, w * h)
    ctx = P.gpu_context(flat)
    ctx.set_traversal("wide")
    for rows, cols in [((h // 3, h), None), (None, (1, 3))]:
        ctx.set_pool(0)
        a = P.gpu_render(ctx, flat, params, rows=rows, cols=cols, rngs=rngs)
        ctx.set_pool(1, 16, 128, 8, 64)
        b = P.gpu_render(ctx, flat, params, rows=rows, cols=cols, rngs=rngs)
        assert _same_state(a, b)
    ctx.close()


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
@pytest.mark.parametrize("pool", [0, 1])
def test_cancel_inside_a_batch(pool):
    """A batch that would run for seconds (the instanced scene, 1920x1080 x 2048 spp) is
    cancelled through the caller's stop flag: the call returns YTHIP_ERR_CANCELLED within
    50 ms of the flag going up, state.samples does not advance, every pixel has taken
    between 0 and `batch` samples, and the context renders correctly afterwards."""
    import scenes as ysc
    flat = ysc.instanced_scene()
    ctx = P.gpu_context(flat)
    ctx.set_pool(pool)
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
    # the next batch lowers the flag again and renders as a fresh context does
    p2 = yt.trace_params(sampler="path", resolution=160, samples=2, batch=2)
    a = P.gpu_render(ctx, flat, p2)
    ctx.close()
    ctx2 = P.gpu_context(flat)
    ctx2.set_pool(pool)
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


# ---- NEE samplers: deferred walks (default) vs the inline kernels -------------------------------
@pytest.mark.parametrize("sampler", ["pathdirect", "pathmis"])
@pytest.mark.parametrize("scene", ["cornellbox", "materials", "lines_points"])
def test_deferred_nee_equals_inline_nee(sampler, scene, monkeypatch):
    """pathdirect / pathmis run their light-pdf walks, NEE rays and NEE emission in the walk
    stage of k_trace (LP_DEFER) by default; YTHIP_NEE_INLINE=1 selects the kernels that do
    them inside the shade step.  Same draws in the same order, same sums: identical state
    (and both equal the reference: tests/test_gpu_parity.py runs the default)."""
    flat = P.SCENES[scene]()
    p = yt.trace_params(sampler=sampler, resolution=160, samples=6, batch=3)
    out = []
    for inline in ["0", "1"]:
        monkeypatch.setenv("YTHIP_NEE_INLINE", inline)
        ctx = P.gpu_context(flat)
        out.append(P.gpu_render(ctx, flat, p))
        ctx.close()
    P.assert_identical(out[0], out[1], f"{sampler} {scene}")


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
