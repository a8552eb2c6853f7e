"""The pixel pool (`-m gpu`; yt_kernels.h resolve_step / DState::pool_next, ythip_set_pixel_pool): a batch launched
as fewer workgroups than tiles whose lanes take the next pixel of a queue when their own has had its samples.
Which lane traces a pixel's samples must not be observable: forced on with a handful of workgroups (so that every
workgroup works through many tiles and the lanes of a wavefront drift far apart), every sampler family on every test
scene gives the reference's whole trace_state, byte for byte — progressive batches, slices, a trace_sample in
between, the measured (default) mode, and a cancelled batch."""
import threading
import time

import numpy as np
import pytest

import parity as P
from parity import yt

pytestmark = pytest.mark.gpu

CASES = [("cornellbox", "path"), ("cornellbox", "pathdirect"), ("cornellbox", "pathmis"), ("materials", "path"), ("materials", "pathmis"),
         ("materials", "naive"), ("lines_points", "path"), ("lines_points", "pathdirect"), ("instances", "path"), ("instances", "eyelight"),
         ("materials", "falsecolor"), ("plane", "pathtest"), ("cornellbox", "furnace")]


@pytest.mark.parametrize("scene,sampler", CASES)
def test_forced_pool_equals_the_reference(scene, sampler):
    flat = P.SCENES[scene]()
    params = yt.trace_params(sampler=sampler, resolution=120, samples=6, batch=3)
    if P.have_ref():
        want = P.RefBundle(flat).render(params)
    else:  # (oracle/_ref did not travel: the plain launch, itself checked against the golden fixtures elsewhere)
        ref = P.gpu_context(flat)
        ref.set_pixel_pool(0)
        want = P.gpu_render(ref, flat, params)
        ref.close()
    for workgroups in (3, 16):
        ctx = P.gpu_context(flat)
        ctx.set_pixel_pool(2, workgroups)
        got = P.gpu_render(ctx, flat, params)
        assert ctx.pixel_pool_info()["on"] == 1
        ctx.close()
        P.assert_identical(want, got, f"{scene} {sampler} pool x{workgroups}")


def test_pool_with_slices_single_samples_and_mode_changes():
    flat = P.SCENES["materials"]()
    p = yt.trace_params(sampler="path", resolution=200, samples=8, batch=2)
    out = []
    for mode in (2, 0):
        ctx = P.gpu_context(flat)
        ctx.set_pixel_pool(mode, 5)
        ctx.make_trace_state(flat, p)
        ctx.trace_samples(p)
        ctx.trace_samples(p)
        ctx.trace_sample(p, 17, 5, 4)
        if mode:
            ctx.set_pixel_pool(0)  # (a plain batch in the middle of a pool render)
        ctx.trace_samples(p)
        if mode:
            ctx.set_pixel_pool(2, 7)
        ctx.trace_samples(p)
        full = ctx.download_state()
        cols = P.gpu_render(ctx, flat, p, cols=(1, 3))
        rows = P.gpu_render(ctx, flat, p, rows=(8, 40))
        out.append((full, cols, rows))
        ctx.close()
    for k, what in enumerate(("full frame", "column slice", "row slice")):
        P.assert_identical(out[0][k], out[1][k], what)


def test_measured_mode_decides_and_changes_nothing():
    """Mode 1 on a frame with more tiles than workgroups: the first batch records the tile costs, the second is the
    timed plain batch, the third the timed pool batch, the fourth finds both finished and decides; the render equals
    the never-pooled one whatever it decided."""
    flat = P.SCENES["lines_points"]()
    p = yt.trace_params(sampler="path", resolution=256, samples=48, batch=8)
    states = []
    for mode in (1, 0):
        ctx = P.gpu_context(flat)
        ctx.set_pixel_pool(mode, 6)
        states.append(P.gpu_render(ctx, flat, p))
        info = ctx.pixel_pool_info()
        ctx.close()
        if mode == 1:
            assert info["decided"] == 1 and info["plain_ms_per_sample"] > 0 and info["pool_ms_per_sample"] > 0, info
        else:
            assert info["decided"] == 0 and info["on"] == 0
    P.assert_identical(states[1], states[0], "measured mode")


def test_cancel_inside_a_pool_batch():
    import scenes as ysc
    flat = ysc.instanced_scene()
    ctx = P.gpu_context(flat)
    ctx.set_pixel_pool(2)
    params = yt.trace_params(sampler="path", resolution=1920, samples=1 << 20, batch=2048)
    warm = yt.trace_params(sampler="path", resolution=1920, samples=1 << 20, batch=1)
    ctx.make_trace_state(flat, warm)
    ctx.trace_samples(warm)  # (records the tile costs: the pool queue is ordered by them)
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
    assert (t_back - t_raise[0]) < 0.05, f"returned {1e3 * (t_back - t_raise[0]):.1f} ms after the flag"
    st = ctx.download_state()
    assert st["samples"] == 1 and st["hits"].max() <= 1 + params.batch and np.isfinite(st["image"]).all()
    assert st["hits"].min() >= 1 and st["hits"].min() < 1 + params.batch  # (some pixel was stopped short of its batch)
    ctx.close()
