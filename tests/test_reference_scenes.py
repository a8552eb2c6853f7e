"""The reference's OWN test corpus (tests/_version43 + tests/_renderings):
scenes read by the reference's own loader (fixtures made by
tests/golden/make_scene_fixtures.py) through the HIP path, against
  (a) the live compiled reference on the same arrays — hit records and trees bit for
      bit, renders within the float tolerance written below;
  (b) the reference's golden renders (converged images of unknown spp/seed, so the
      match is statistical: image mean and 8x8-block means).
These scenes exercise what the synthetic ones only sample: lat-long environment maps
with importance sampling, sRGB / normal-map / roughness textures, every material type,
area lights, instancing, quads with degenerate corners."""
import os

import numpy as np
import pytest

import parity as P
from parity import ry, yt

NAMES = P.ref_scene_names()
needs_ref = pytest.mark.skipif(not P.have_ref(), reason="oracle/_ref not built / did not travel")
REF_TESTS = "/root/reference/tests"


def test_fixtures_present():
    assert {"features1", "materials1", "materials2", "materials3", "materials4", "shapes1",
            "instances1"} <= set(NAMES)
    # round 6: every scene of tests/_version43 the reference's own loader can read (features2 / shapes2 / shapes3 name .ply
    # files that are not upstream) — 14 of 17
    assert {"shapes4", "materials5", "environments2", "furnace1", "arealights1", "environments1", "furnace2"} <= set(NAMES)


@needs_ref
@pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="reference corpus not present (GPU box)")
@pytest.mark.parametrize("name", ["features1", "materials3", "shapes1", "furnace2", "shapes4", "materials5", "environments2", "furnace1"])
def test_fixtures_are_what_the_reference_loader_reads(name):
    flat = P.load_ref_scene(name)
    ref = ry.RefScene.load(f"{REF_TESTS}/_version43/{name}/{name}.json").flat()
    for f in P._FIELDS:
        assert np.asarray(getattr(flat, f)).tobytes() == np.asarray(getattr(ref, f)).tobytes(), f


@needs_ref
def test_reference_reproduces_its_golden_render_statistically():
    """Calibrates the statistical bar of the GPU test below with the reference itself:
    16 spp of features1 at the golden's size vs the golden's block means."""
    flat = P.load_ref_scene("features1")
    g = np.load(os.path.join(P.REF_SCENE_DIR, "golden_features1.npz"))
    p = yt.trace_params(sampler="path", resolution=1280, samples=16, batch=16)
    st = P.RefBundle(flat).render(p)
    assert (st["height"], st["width"]) == tuple(g["shape"])
    img = st["image"][:, :3].reshape(st["height"], st["width"], 3)
    assert abs(img.mean() - g["mean"]) <= 0.015 * g["mean"]
    # block level: the reference built today sits at 0.14 on this measure at 16 AND at 64
    # spp — the goldens were rendered by an older revision of scenes / textures; they pin
    # the image statistically, not locally.  The GPU test uses the same bar.
    h, w = img.shape[0] // 8 * 8, img.shape[1] // 8 * 8
    blocks = img[:h, :w].reshape(h // 8, 8, w // 8, 8, 3).mean((1, 3))
    assert np.abs(blocks - g["blocks"]).mean() / g["mean"] <= 0.16


# ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gpu_bundles():
    cache = {}

    def get(name):
        if name not in cache:
            flat = P.load_ref_scene(name)
            cache[name] = (flat, P.gpu_context(flat), P.RefBundle(flat) if P.have_ref() else None)
        return cache[name]

    yield get
    for _, ctx, _ in cache.values():
        ctx.close()


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("name", NAMES)
def test_hits_and_tree_bit_exact_vs_reference(gpu_bundles, name):
    flat, ctx, rb = gpu_bundles(name)
    ref_bvh = rb.bvh.flat()
    mine = ctx.download_bvh()
    assert mine.nodes.tobytes() == ref_bvh.nodes.tobytes() and mine.primitives.tobytes() == ref_bvh.primitives.tobytes()
    rays = P.random_rays(flat, 100000, seed=31)
    assert P.hits_equal(ctx.intersect_batch(rays), ry.intersect_batch(rb.bvh, rb.scene, rays))
    p = yt.trace_params(resolution=640, samples=1)
    ctx.make_trace_state(flat, p)
    cam = ctx.camera_rays(p)
    st = ry.RefState(rb.scene, p)
    assert cam.tobytes() == ry.camera_rays(st, rb.scene, p).tobytes()
    assert P.hits_equal(ctx.intersect_batch(cam), ry.intersect_batch(rb.bvh, rb.scene, cam))


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("name", NAMES)
def test_render_vs_live_reference(gpu_bundles, name):
    """`path`, default params, 320 px, 4 spp: the whole trace_state bit for bit (lat-long
    environment maps with importance sampling, sRGB / normal-map / roughness textures,
    every material type, area lights, instancing — all through glibc's libm as restated
    in yt_libm.h)."""
    flat, ctx, rb = gpu_bundles(name)
    sampler = "furnace" if name.startswith("furnace") else "path"
    p = yt.trace_params(sampler=sampler, resolution=320, samples=4, batch=2)
    gpu = P.gpu_render(ctx, flat, p)
    ref = rb.render(p)
    assert np.isfinite(gpu["image"]).all()
    P.assert_identical(gpu, ref, name)


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("sampler", ["pathdirect", "pathmis", "naive", "eyelight", "falsecolor"])
def test_other_samplers_on_features1(gpu_bundles, sampler):
    """The other integrators on the reference's feature scene (textures, environment
    map, area lights, every material): bit for bit, like `path` above."""
    flat, ctx, rb = gpu_bundles("features1")
    p = yt.trace_params(sampler=sampler, resolution=256, samples=2, batch=2, falsecolor="normal")
    gpu = P.gpu_render(ctx, flat, p)
    ref = rb.render(p)
    P.assert_identical(gpu, ref, f"features1 {sampler}")


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("name", ["features1", "materials2"])
def test_converged_render_vs_reference(gpu_bundles, name):
    """Converged images, independent rng streams (GPU seed 7, reference default seed):
    640 px, GPU 256 spp vs reference 128 spp — image mean within 0.5 %, mean absolute
    error of the 8x8-block means <= 3 % of the image mean (Monte Carlo noise of the
    128 spp side)."""
    flat, ctx, rb = gpu_bundles(name)
    gpu = P.gpu_render(ctx, flat, yt.trace_params(sampler="path", resolution=640, samples=256, batch=64, seed=7))
    ref = rb.render(yt.trace_params(sampler="path", resolution=640, samples=128, batch=64))
    h, w = gpu["height"], gpu["width"]

    def blocks(st):
        img = st["image"][:, :3].reshape(h, w, 3)
        return img[:h // 8 * 8, :w // 8 * 8].reshape(h // 8, 8, w // 8, 8, 3).mean((1, 3))
    a, b = blocks(gpu), blocks(ref)
    assert abs(a.mean() - b.mean()) <= 0.005 * b.mean(), (a.mean(), b.mean())
    assert np.abs(a - b).mean() <= 0.03 * b.mean(), np.abs(a - b).mean() / b.mean()


# materials4's golden belongs to another revision of the scene (the reference built today is
# 47 % off in the mean), so it pins nothing and is left out.
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["features1", "materials1", "materials2", "materials3"])
def test_gpu_render_matches_the_references_golden_render(gpu_bundles, name):
    """The reference's golden HDRs (tests/_renderings/<name>-mst.hdr, 1280x533,
    converged, spp / seed / revision unknown — SURVEY.md §4): a 256 spp GPU render
    agrees with them as well as the reference built today does: image mean within 3 %
    (the reference at 32 spp: 0.7 / 1.8 / 2.4 / 2.2 % for the four scenes), mean
    absolute error of the 8x8-block means <= 17 % of the image mean (the reference:
    14 / 14 / 10 / 16 % — the goldens come from an older revision of scenes and
    textures, they pin the images statistically, not locally)."""
    flat, ctx, _ = gpu_bundles(name)
    g = np.load(os.path.join(P.REF_SCENE_DIR, f"golden_{name}.npz"))
    p = yt.trace_params(sampler="path", resolution=1280, samples=256, batch=64)
    st = P.gpu_render(ctx, flat, p)
    assert (st["height"], st["width"]) == tuple(g["shape"])
    img = st["image"][:, :3].reshape(st["height"], st["width"], 3)
    h, w = img.shape[0] // 8 * 8, img.shape[1] // 8 * 8
    blocks = img[:h, :w].reshape(h // 8, 8, w // 8, 8, 3).mean((1, 3))
    mean_err = abs(img.mean() - g["mean"]) / g["mean"]
    block_err = np.abs(blocks - g["blocks"]).mean() / g["mean"]
    print(name, "mean", float(img.mean()), "golden", float(g["mean"]), "mean_err", float(mean_err),
          "block_err", float(block_err))
    assert mean_err <= 0.03, mean_err
    assert block_err <= 0.17, block_err
