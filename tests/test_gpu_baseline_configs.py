"""BASELINE.json's configs at FULL size (`-m gpu`): cfg2 (1M-triangle plane), cfg2b
(the Cornell box with 1M-triangle walls: deep paths, area-light pdf walks), cfg4
(10,000 instances of a 1,024-triangle sphere) and cfg5 (800,000 hair segments,
subsurface) — bit-exact hit records for whole frames of camera rays against the
live compiled reference, renders equal to the reference's whole trace_state byte for byte, and
size-independent properties (row shards == full frame) where no reference run
is needed."""
import numpy as np
import pytest

import parity as P
from parity import ry, yt, ysc

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not P.have_ref(), reason="oracle/_ref did not travel")


def hair_scene():
    """cfg5: make_hair(make_sphere(32, 1), {8, 100000}, {0.2, 0.2},
    {0.002, 0.001}) generated ONCE by the g++ reference (its sample_shape call has
    unsequenced rng arguments, so the geometry is compiler-dependent) and fed as
    the same arrays to both sides."""
    rs = ry.RefScene.new()
    # (SURVEY.md §8d gives the geometry, materials, environment, frame size and spp of cfg5 but no camera; its probe popped
    #  193.7 nodes per sample.  From cfg2's camera (0, 3, 8) the ball covers a twentieth of the frame — 46 nodes per sample —,
    #  from here 290: the survey's probe stood about 3.9 units away.  This camera is the one every round has tested with.)
    rs.add_camera_lookat((0, 0.5, 3.2), (0, 0, 0), lens=0.035, film=0.036, aspect=16 / 9)
    base = rs.add_sphere(32, 1.0)
    hair = rs.add_hair(base, (8, 100000), (0.2, 0.2), (0.002, 0.001))
    m_base = rs.add_material(type="matte", color=(0.7, 0.7, 0.7))
    m_hair = rs.add_material(type="subsurface", color=(0.8, 0.6, 0.4), roughness=0.3,
                             scattering=(0.5, 0.5, 0.5))
    rs.add_instance(base, m_base)
    rs.add_instance(hair, m_hair)
    rs.add_environment((1, 1, 1))
    return rs.flat()


@pytest.fixture(scope="module")
def cfg(request):
    cache = {}

    def get(name):
        if name not in cache:
            flat = {"cfg2": ysc.plane_scene, "cfg2b": P.scene_cornell_1m, "cfg4": ysc.instanced_scene,
                    "cfg5": hair_scene}[name]()
            cache[name] = (flat, P.gpu_context(flat), P.RefBundle(flat) if P.have_ref() else None)
        return cache[name]

    yield get
    for _, ctx, _ in cache.values():
        ctx.close()


def test_cfg2_is_the_baseline_scene(cfg):
    flat, ctx, _ = cfg("cfg2")
    assert int(flat.shapes[0]["num_triangles"]) == 1_000_000
    assert int(flat.shapes[0]["num_positions"]) == 501_501
    b = ctx.download_bvh()
    assert len(b.nodes) == 503_478  # 503,478 nodes (BLAS + the 1-node TLAS)


@needs_ref
@pytest.mark.parametrize("name,res", [("cfg2", 1280), ("cfg2b", 1024), ("cfg4", 1920), ("cfg5", 1280)])
def test_full_frame_primary_hits_bit_exact(cfg, name, res):
    """Every primary ray of the BASELINE frame (0.9M / 2M rays): identical
    (instance, element, uv, distance) to intersect_scene_bvh of the reference."""
    flat, ctx, rb = cfg(name)
    p = yt.trace_params(resolution=res, samples=1)
    ctx.make_trace_state(flat, p)
    rays = ctx.camera_rays(p)
    st = ry.RefState(rb.scene, p)
    assert rays.tobytes() == ry.camera_rays(st, rb.scene, p).tobytes()
    a = ctx.intersect_batch(rays)
    b = ry.intersect_batch(rb.bvh, rb.scene, rays)
    assert P.hits_equal(a, b)
    assert a["hit"].mean() > 0.2
    # incoherent rays too, incl. zero-component / axis-aligned directions
    rr = P.random_rays(flat, 200_000, seed=23)
    assert P.hits_equal(ctx.intersect_batch(rr), ry.intersect_batch(rb.bvh, rb.scene, rr))


@needs_ref
@pytest.mark.parametrize("name,res,spp", [("cfg2", 1280, 2), ("cfg2b", 256, 2), ("cfg4", 640, 4), ("cfg5", 320, 2)])
def test_render_vs_live_reference(cfg, name, res, spp):
    """`path`, 8 bounces, clamp 10, default seed, reduced size: the whole trace_state bit
    for bit (no float tolerance left: the device evaluates the reference platform's libm)."""
    flat, ctx, rb = cfg(name)
    params = yt.trace_params(sampler="path", resolution=res, samples=spp, batch=spp)
    gpu = P.gpu_render(ctx, flat, params)
    ref = rb.render(params)
    assert np.isfinite(gpu["image"]).all()
    P.assert_identical(gpu, ref, name)


@needs_ref
@pytest.mark.parametrize("name,res,spp", [("cfg2", 1280, 64), ("cfg2b", 1024, 64), ("cfg4", 1920, 256),
                                          ("cfg5", 1280, 64)])
def test_render_at_baseline_size_vs_live_reference(cfg, name, res, spp):
    """The BASELINE configs AT THEIR FULL SIZE — configs[1] 1280x720x64 spp, cfg2b
    1024x1024x64, configs[3] 1920x1080x256, configs[4] 1280x720x64; `path`, 8 bounces, clamp
    10, default seed — against the live reference (a few seconds to a minute of its 256
    threads each): every pixel's 64 / 256-sample sequential chain — its rng stream, the
    russian roulette after bounce 3, the running means and the hit counter — bit for bit;
    one batch here, two half batches for configs[1] (progressive == one-shot)."""
    flat, ctx, rb = cfg(name)
    params = yt.trace_params(sampler="path", resolution=res, samples=spp, batch=spp // 2 if name == "cfg2" else spp)
    gpu = P.gpu_render(ctx, flat, params)
    # (shared with tests/test_gpu_own_tree.py: one CPU render per config and session)
    key = ({"cfg2": "cfg2", "cfg2b": "cfg2b", "cfg4": "cfg4", "cfg5": "cfg5_testcam"}[name], res, spp)
    if key not in P.REF_RENDERS:
        P.REF_RENDERS[key] = rb.render(params)
    ref = P.REF_RENDERS[key]
    assert gpu["samples"] == ref["samples"] == spp
    assert int(gpu["hits"].max()) == spp
    P.assert_identical(gpu, ref, f"{name} at {res} x {spp} spp")
    # ... and the same frame on the streaming scheduler (csrc/yt_stream.h: every pixel in flight, sorted extend stage, tail kernel),
    # as two half batches: the same bytes
    ctx.set_scheduler(1)
    try:
        half = yt.trace_params(sampler="path", resolution=res, samples=spp, batch=spp // 2)
        streamed = P.gpu_render(ctx, flat, half)
        assert ctx.stream_info()["ran"] == 1
    finally:
        ctx.set_scheduler(2)
    P.assert_identical(streamed, ref, f"{name} at {res} x {spp} spp, streamed")


def test_cfg2_full_size_properties(cfg):
    """No reference needed: (1) rows [0,90) and [630,720) rendered as slices equal
    the same rows of the full 1280x720x4 frame bit for bit (§8e sharding);
    (2) two batches of 2 equal one batch of 4 (progressive == one-shot);
    (3) the image mean sits at the survey's 0.748 +- 2 %."""
    flat, ctx, _ = cfg("cfg2")
    params = yt.trace_params(sampler="path", resolution=1280, samples=4, batch=4)
    full = P.gpu_render(ctx, flat, params)
    w, h = full["width"], full["height"]
    assert (w, h) == (1280, 720)
    rngs = yt.make_rngs(params.seed, w * h)
    for r0, r1 in [(0, 90), (630, 720), (355, 361)]:
        part = P.gpu_render(ctx, flat, params, rows=(r0, r1), rngs=rngs)
        for k in ["image", "albedo", "normal", "hits", "rngs"]:
            assert part[k].tobytes() == full[k][r0 * w:r1 * w].tobytes(), (k, r0, r1)
    import sharding
    sh = sharding.shard_frame(w, h, 8, 5)  # what rank 5 of 8 renders in bench.py
    part = P.gpu_render(ctx, flat, params, rows=sh.rows, cols=sh.cols, rngs=rngs)
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert part[k].tobytes() == full[k][sh.pixels].tobytes(), (k, "columns 5/8")
    prog = P.gpu_render(ctx, flat, yt.trace_params(sampler="path", resolution=1280, samples=4, batch=2))
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert prog[k].tobytes() == full[k].tobytes(), k
    mean = float(full["image"][:, :3].mean())
    assert abs(mean - 0.748) < 0.02 * 0.748, mean
