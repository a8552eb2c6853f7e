"""The streaming scheduler (`-m gpu`; csrc/yt_stream.h, ythip_set_scheduler): every pixel in flight, SoA path state in
HBM, per bounce a counting sort of the next rays + a traversal-only extend kernel + a shade kernel in pixel order.
It schedules the same per-pixel operations as the fused kernel, so the whole trace_state must be the reference's,
byte for byte — on every scene class (matte triangles, no-texture, opaque textured, general; with and without area
lights), for every sort order incl. the unsorted one, progressive batches, slices, cancellation — and a sampler or
mode it does not serve must quietly run on the fused kernel."""
import threading
import time

import numpy as np
import pytest

import parity as P
from parity import yt

pytestmark = pytest.mark.gpu

SCENES = ["cornellbox", "plane", "materials", "instances", "lines_points"]


def stream_context(flat):
    """A context on the streaming scheduler.  The test scenes' trees are tiny (< 64 primitives: the library would walk them
    binary and the scheduler, which has the wide walk only, would decline), so the wide walk is forced."""
    ctx = P.gpu_context(flat)
    ctx.set_traversal("wide")
    ctx.set_scheduler(1)
    return ctx


def want_state(flat, params):
    if P.have_ref():
        return P.RefBundle(flat).render(params)
    ref = P.gpu_context(flat)  # (oracle/_ref did not travel: the fused kernel, itself checked against the fixtures elsewhere)
    want = P.gpu_render(ref, flat, params)
    ref.close()
    return want


@pytest.mark.parametrize("scene", SCENES)
def test_streamed_path_equals_the_reference(scene):
    flat = P.SCENES[scene]()
    params = yt.trace_params(sampler="path", resolution=144, samples=12, batch=6)
    want = want_state(flat, params)
    for order, cells in ((0, 4), (1, 3), (2, 4)):
        ctx = stream_context(flat)
        ctx.set_stream_options(order=order, cell_bits=cells)
        got = P.gpu_render(ctx, flat, params)
        info = ctx.stream_info()
        ctx.close()
        assert info["ran"] == 1 and info["generations"] >= params.batch, info
        P.assert_identical(want, got, f"{scene} streamed, order {order}, {cells} cell bits")


@pytest.mark.parametrize("scene", ["materials", "instances", "cornellbox"])
def test_groups_change_nothing(scene, monkeypatch):
    """One, two, five chains of generations on as many streams: the same bytes, over several batches."""
    monkeypatch.setenv("YTHIP_STREAM_MIN_SLOTS", "1024")
    flat = P.SCENES[scene]()
    params = yt.trace_params(sampler="path", resolution=200, samples=12, batch=4)
    want = want_state(flat, params)
    for groups in (1, 2, 5):
        ctx = stream_context(flat)
        ctx.set_stream_groups(groups)
        got = P.gpu_render(ctx, flat, params)
        info = ctx.stream_info()
        ctx.close()
        assert info["ran"] == 1 and info["groups"] == groups, info
        P.assert_identical(want, got, f"{scene} streamed, {groups} groups")


@pytest.mark.parametrize("scene", SCENES)
def test_the_tail_kernel_changes_nothing(scene, monkeypatch):
    """ythip_set_stream_finish: the paths still queued when a group's queue has shrunk to a fraction of its slots finish their
    pixels' batch in ONE launch (ks_finish: extend and shade in turn, a lane per queue entry) instead of further generations —
    never (0), late, early, and from the first ray on (1000: the whole batch in ks_finish): the reference's bytes each time."""
    monkeypatch.setenv("YTHIP_STREAM_MIN_SLOTS", "1024")
    flat = P.SCENES[scene]()
    params = yt.trace_params(sampler="path", resolution=160, samples=12, batch=6)
    want = want_state(flat, params)
    for permille, groups in ((0, 2), (250, 1), (250, 2), (700, 2), (1000, 1), (1000, 3)):
        ctx = stream_context(flat)
        ctx.set_stream_finish(permille)
        ctx.set_stream_groups(groups)
        got = P.gpu_render(ctx, flat, params)
        info = ctx.stream_info()
        ctx.close()
        assert info["ran"] == 1 and info["groups"] == groups, info
        if permille == 0:
            assert info["finish_rays"] == 0, info
        if permille >= 700 and scene != "plane":  # (the open plane's queue is full or empty: every sample is one generation)
            assert info["finish_rays"] > 0, info
        if permille == 1000:
            assert info["generations"] <= 1 and info["finish_rays"] == info["path_slots"], info  # (the scan behind ks_init counts as one)
        P.assert_identical(want, got, f"{scene} streamed, finish at {permille} / 1000, {groups} groups")


@pytest.mark.parametrize("scene", SCENES)
def test_streamed_pathdirect_equals_the_reference(scene, monkeypatch):
    """Sampler `pathdirect` (yocto_trace.cpp:599-722) on the streaming scheduler: the path rays through the sorted extend stage, the
    NEE half of a bounce — light pdf walks, the NEE ray, its emission — in the shade stage, as k_trace's deferred stage does it.
    The reference's bytes, with and without the tail kernel, in every mode's unit."""
    monkeypatch.setenv("YTHIP_STREAM_MIN_SLOTS", "1024")
    flat = P.SCENES[scene]()
    params = yt.trace_params(sampler="pathdirect", resolution=144, samples=12, batch=6)
    want = want_state(flat, params)
    for order, groups, permille in ((0, 2, 250), (2, 1, 0), (1, 3, 1000)):
        ctx = stream_context(flat)
        ctx.set_stream_options(order=order)
        ctx.set_stream_groups(groups)
        ctx.set_stream_finish(permille)
        got = P.gpu_render(ctx, flat, params)
        assert ctx.stream_info()["ran"] == 1
        ctx.close()
        P.assert_identical(want, got, f"{scene} pathdirect streamed, order {order}, {groups} chains, tail kernel at {permille}")


@pytest.mark.parametrize("scene", SCENES)
@pytest.mark.parametrize("sampler", ["naive", "pathtest"])
def test_streamed_naive_and_pathtest_equal_the_reference(scene, sampler, monkeypatch):
    """`naive` (yocto_trace.cpp:1032-1108) and `pathtest` (:937-1029) on the streaming scheduler — general-class shade kernels, as their
    fused kernels are —: the reference's bytes (the tolerance / own-tree modes: test_streamed_nee_naive_pathtest_in_the_other_modes)."""
    monkeypatch.setenv("YTHIP_STREAM_MIN_SLOTS", "1024")
    flat = P.SCENES[scene]()
    params = yt.trace_params(sampler=sampler, resolution=144, samples=12, batch=6)
    want = want_state(flat, params)
    for groups, permille in ((2, 250), (1, 1000)):
        ctx = stream_context(flat)
        ctx.set_stream_groups(groups)
        ctx.set_stream_finish(permille)
        got = P.gpu_render(ctx, flat, params)
        assert ctx.stream_info()["ran"] == 1
        ctx.close()
        P.assert_identical(want, got, f"{scene} {sampler} streamed, {groups} chains, tail kernel at {permille}")


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("sampler", ["pathdirect", "naive", "pathtest"])
def test_streamed_nee_naive_pathtest_in_the_other_modes(mode, sampler):
    flat = P.SCENES["materials"]()
    params = yt.trace_params(sampler=sampler, resolution=128, samples=8, batch=8, fastmath=mode)
    out = []
    for stream in (0, 1):
        ctx = P.gpu_context(flat)
        ctx.set_traversal("wide")
        if mode == 2:
            ctx.make_own_bvh(flat)
        ctx.set_scheduler(stream)
        out.append(P.gpu_render(ctx, flat, params))
        assert ctx.stream_info()["ran"] == stream and ctx.last_launch_fastmath() == mode
        ctx.close()
    P.assert_identical(out[0], out[1], f"{sampler} fastmath {mode}: streamed vs fused")


@pytest.mark.parametrize("mode", [1, 2])
def test_the_tail_kernel_in_the_other_modes(mode):
    flat = P.SCENES["materials"]()
    params = yt.trace_params(sampler="path", resolution=144, samples=8, batch=8, fastmath=mode)
    out = []
    for stream, permille in ((0, 0), (1, 500), (1, 1000)):
        ctx = P.gpu_context(flat)
        ctx.set_traversal("wide")
        if mode == 2:
            ctx.make_own_bvh(flat)
        ctx.set_scheduler(stream)
        ctx.set_stream_finish(permille)
        out.append(P.gpu_render(ctx, flat, params))
        assert ctx.stream_info()["ran"] == stream and ctx.last_launch_fastmath() == mode
        if permille == 1000:
            assert ctx.stream_info()["finish_rays"] > 0
        ctx.close()
    P.assert_identical(out[0], out[1], f"fastmath {mode}: streamed with a tail kernel vs fused")
    P.assert_identical(out[0], out[2], f"fastmath {mode}: the whole batch in the tail kernel vs fused")


@pytest.mark.parametrize("kw", [dict(tentfilter=True), dict(nocaustics=True), dict(envhidden=True), dict(bounces=1), dict(bounces=3, clamp=2.0)])
def test_streamed_params_variants(kw):
    flat = P.SCENES["materials"]()
    params = yt.trace_params(sampler="path", resolution=96, samples=8, batch=8, **kw)
    want = want_state(flat, params)
    ctx = stream_context(flat)
    got = P.gpu_render(ctx, flat, params)
    assert ctx.stream_info()["ran"] == 1
    ctx.close()
    P.assert_identical(want, got, f"materials streamed {kw}")


def test_general_class_and_specialised_classes_agree():
    """ythip_set_specialization(0) streams the general-class kernels on a matte scene: the same bytes."""
    flat = P.SCENES["cornellbox"]()
    params = yt.trace_params(sampler="path", resolution=128, samples=8, batch=8)
    out = []
    for spec in (1, 0):
        ctx = stream_context(flat)
        ctx.set_specialization(spec)
        out.append(P.gpu_render(ctx, flat, params))
        assert ctx.stream_info()["ran"] == 1
        ctx.close()
    P.assert_identical(out[0], out[1], "class 1 vs class 0, streamed")


def test_slices_batches_and_scheduler_changes_in_one_render():
    flat = P.SCENES["instances"]()
    p = yt.trace_params(sampler="path", resolution=160, samples=16, batch=4)
    out = []
    for stream in (1, 0):
        ctx = P.gpu_context(flat)
        ctx.set_traversal("wide")
        ctx.set_scheduler(stream)
        ctx.make_trace_state(flat, p)
        ctx.trace_samples(p)
        ctx.trace_sample(p, 9, 3, 4)  # (a single sample in between: always the fused kernel)
        if stream:
            ctx.set_scheduler(0)  # a fused batch in the middle of a streamed render
        ctx.trace_samples(p)
        if stream:
            ctx.set_scheduler(1)
        ctx.trace_samples(p)
        ctx.trace_samples(p)
        full = ctx.download_state()
        cols = P.gpu_render(ctx, flat, p, cols=(1, 3))
        rows = P.gpu_render(ctx, flat, p, rows=(8, 40))
        out.append((full, cols, rows))
        ctx.close()
    for k, what in enumerate(("full frame", "column slice", "row slice")):
        P.assert_identical(out[0][k], out[1][k], what)


def test_what_the_scheduler_does_not_serve_runs_fused():
    flat = P.SCENES["cornellbox"]()
    ctx = stream_context(flat)
    for kw in (dict(sampler="furnace", batch=4), dict(sampler="path", batch=1), dict(sampler="path", batch=4, bounces=0),
               dict(sampler="pathmis", batch=4, fastmath=1)):
        params = yt.trace_params(resolution=64, samples=4, **kw)
        got = P.gpu_render(ctx, flat, params)
        assert ctx.stream_info()["ran"] == 0, kw
        if not kw.get("fastmath"):
            P.assert_identical(want_state(flat, params), got, f"fused fallback {kw}")
    ctx.set_traversal("binary")  # the binary walk: not served either
    params = yt.trace_params(sampler="path", resolution=64, samples=4, batch=4)
    P.assert_identical(want_state(flat, params), P.gpu_render(ctx, flat, params), "fused fallback, binary walk")
    assert ctx.stream_info()["ran"] == 0
    ctx.close()


@pytest.mark.parametrize("scene", SCENES)
@pytest.mark.parametrize("mode", [1, 2])
def test_streamed_tolerance_and_own_tree_modes_equal_their_fused_kernels(scene, mode):
    """fastmath = 1 / 2 on the streaming scheduler: the tolerance and the own-tree unit carry their own build of the stream
    kernels (csrc/yt_stream_unit.h).  The scheduler changes which wavefront walks a ray, never what a pixel computes, so a streamed
    batch equals the FUSED batch of the same mode byte for byte (what those modes are gated against is the reference, elsewhere:
    tests/test_gpu_fastmath.py, test_gpu_own_tree.py)."""
    flat = P.SCENES[scene]()
    params = yt.trace_params(sampler="path", resolution=144, samples=12, batch=6, fastmath=mode)
    out = []
    for stream in (0, 1):
        ctx = P.gpu_context(flat)
        ctx.set_traversal("wide")
        if mode == 2:
            ctx.make_own_bvh(flat)
        ctx.set_scheduler(stream)
        out.append(P.gpu_render(ctx, flat, params))
        assert ctx.stream_info()["ran"] == stream and ctx.last_launch_fastmath() == mode, (ctx.stream_info(), ctx.last_launch_fastmath())
        ctx.close()
    P.assert_identical(out[0], out[1], f"{scene} fastmath {mode}: streamed vs fused")


def test_the_measured_choice_of_scheduler():
    """ythip_set_scheduler(ctx, 2): once the fused path has settled one batch is timed fused, the next two run streamed (the second timed), and the faster
    serves the state from then on; a new state / batch size is measured again; the render is the reference's whichever ran."""
    flat = P.SCENES["cornellbox"]()
    params = yt.trace_params(sampler="path", resolution=256, samples=96, batch=8)
    want = want_state(flat, params)
    ctx = P.gpu_context(flat)
    ctx.set_traversal("wide")
    ctx.set_scheduler(2)
    ctx.make_trace_state(flat, params)
    ran, states = [], []
    for _ in range(params.samples // params.batch):
        ctx.trace_samples(params)
        info = ctx.stream_info()
        ran.append(info["ran"]), states.append(info["choice_state"])
    got = ctx.download_state()
    P.assert_identical(want, got, "measured choice of scheduler")
    assert states[-1] == 4 and sum(ran) >= 2, (ran, states)  # decided; two batches were the streamed probes (more if streaming won)
    k = states.index(3)  # the timed streamed batch; the one before it streamed too (untimed), the one before that was the timed fused one
    assert ran[k] == 1 and ran[k - 1] == 1 and ran[k - 2] == 0, (ran, states)
    info = ctx.stream_info()
    assert info["fused_ms_per_sample"] > 0 and info["stream_ms_per_sample"] > 0
    assert all(r == info["choice_streamed"] for r in ran[k + 1:]), (ran, info)
    # another batch size: measured again
    q = yt.trace_params(sampler="path", resolution=256, samples=1000, batch=16)
    ctx.trace_samples(q)
    assert ctx.stream_info()["choice_state"] < 4
    # batches too short to time run fused and leave the choice open
    ctx.set_scheduler(2)
    r = yt.trace_params(sampler="path", resolution=256, samples=1000, batch=4)
    ctx.trace_samples(r)
    assert ctx.stream_info()["ran"] == 0 and ctx.stream_info()["choice_state"] == 0
    ctx.close()


def test_the_measured_choice_leaves_the_async_call_alone():
    """A streamed batch is enqueued by a host loop that returns when the batch is done, so under the default (the measured choice)
    only ythip_trace_samples — whose caller waits anyway — is ever streamed: ythip_trace_samples_async keeps returning at once,
    its batches run fused and take no part in the choice.  Mode 1 streams the async call's batches too (it then blocks)."""
    flat = P.SCENES["cornellbox"]()
    p = yt.trace_params(sampler="path", resolution=256, samples=1 << 20, batch=8)
    ctx = P.gpu_context(flat)
    ctx.set_traversal("wide")
    assert ctx.lib.ythip_get_scheduler(ctx.h) == 2
    ctx.make_trace_state(flat, p)
    for _ in range(8):
        ctx.trace_samples_async(p)
        ctx.sync()
        assert ctx.stream_info()["ran"] == 0 and ctx.stream_info()["choice_state"] == 0
    ctx.set_scheduler(1)
    ctx.trace_samples_async(p)
    assert ctx.stream_info()["ran"] == 1
    ctx.sync()
    ctx.close()


def test_own_tree_mode_without_its_tree_fails_on_the_streaming_scheduler_too():
    flat = P.SCENES["cornellbox"]()
    ctx = stream_context(flat)
    params = yt.trace_params(sampler="path", resolution=64, samples=4, batch=4, fastmath=2)
    with pytest.raises(yt.YthipError, match="own tree"):
        P.gpu_render(ctx, flat, params)
    ctx.close()


def test_cancel_inside_a_streamed_batch():
    """The caller's stop flag ends a streamed batch at the pixels' next sample boundaries: the call returns CANCELLED,
    state.samples stays, every pixel holds a whole number of samples (hits counts them) and the next batch runs."""
    flat = P.SCENES["cornellbox"]()
    ctx = stream_context(flat)
    p = yt.trace_params(sampler="path", resolution=512, samples=100000, batch=2048)
    ctx.make_trace_state(flat, p)
    stop = np.zeros(1, np.int32)
    threading.Timer(0.05, lambda: stop.__setitem__(0, 1)).start()
    t0 = time.time()
    with pytest.raises(yt.YthipError, match="cancelled"):
        ctx.trace_samples(p, stop=stop)
    assert time.time() - t0 < 5.0
    st = ctx.download_state()
    assert st["samples"] == 0
    assert np.isfinite(st["image"]).all()
    assert 0 < st["hits"].max() < 2048
    q = yt.trace_params(sampler="path", resolution=512, samples=100000, batch=4)
    ctx.trace_samples(q)
    assert ctx.stream_info()["ran"] == 1
    ctx.close()


def test_cancel_inside_the_tail_kernel():
    """ks_finish polls the stop word once per bounce: a batch that runs entirely inside it stops at sample boundaries too."""
    flat = P.SCENES["cornellbox"]()
    ctx = stream_context(flat)
    ctx.set_stream_finish(1000)
    p = yt.trace_params(sampler="path", resolution=512, samples=100000, batch=4096)
    ctx.make_trace_state(flat, p)
    stop = np.zeros(1, np.int32)
    threading.Timer(0.05, lambda: stop.__setitem__(0, 1)).start()
    t0 = time.time()
    with pytest.raises(yt.YthipError, match="cancelled"):
        ctx.trace_samples(p, stop=stop)
    assert time.time() - t0 < 5.0
    st = ctx.download_state()
    assert st["samples"] == 0 and np.isfinite(st["image"]).all() and 0 < st["hits"].max() < 4096
    ctx.close()


def test_profiling_reports_the_walks_evenness():
    flat = P.SCENES["cornellbox"]()
    ctx = stream_context(flat)
    ctx.set_profiling(1)
    ctx.set_stream_finish(0)  # (every ray through ks_extend, which does the counting)
    p = yt.trace_params(sampler="path", resolution=128, samples=8, batch=8)
    P.gpu_render(ctx, flat, p)
    info = ctx.stream_info()
    gens = ctx.stream_generations()
    ctx.close()
    assert info["rays"] > 128 * 128 * 8 and 0 < info["lane_steps"] <= info["wave_steps"], info
    assert len(gens) == info["generations"] and gens[0] == 128 * 128 and gens.sum() == info["rays"] and (gens > 0).all()
