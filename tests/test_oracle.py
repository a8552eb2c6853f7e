"""Pins the CPU restatement (oracle/yt_oracle.cpp) — CPU only, `-m "not gpu"`:
  (a) against the reference's known answers,
  (b) against the golden fixtures generated from the compiled reference,
  (c) bit for bit against the live compiled reference (oracle/_ref) when present.
Everything here is bit-exact: restatement and reference are both g++ builds on
the same libm, and the restatement keeps the reference's operation order."""
import os

import numpy as np
import pytest

import parity as P
from parity import ry, yt

import ytoracle as yo

needs_ref = pytest.mark.skipif(not P.have_ref(), reason="oracle/_ref not built")
KAT = np.load(os.path.join(P.GOLDEN, "kat.npz"))

# scenes the restatement covers end to end (matte, no textures): BASELINE cfg1 and a
# small cfg2; it intersects every scene
RENDERABLE = {"cornellbox": ["path", "naive", "eyelight"], "plane": ["path", "eyelight"]}


@pytest.fixture(scope="module")
def bundles():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = yo.Bundle(P.SCENES[name]())
        return cache[name]
    return get


def test_pcg_and_seeding_known_answers():
    state, r = yo.rand1f(961748941, 1, 8)
    assert tuple(state) == (17286221497386715027, 3)
    assert np.allclose(r[:4], [0.893633127, 0.246839881, 0.458433747, 0.477094531], atol=1e-9)
    assert np.array_equal(state, KAT["make_rng"]) and np.array_equal(r, KAT["rand1f"])
    rngs = yo.make_rngs(961748941, 256 * 256)
    assert tuple(rngs[0]) == (10473825460370060855, 1450249601)
    assert tuple(rngs[-1]) == (1965839424625278335, 94108541)
    assert np.array_equal(rngs, yt.make_rngs(961748941, 256 * 256))  # == the product's host seeding


def test_state_size_rule():
    cam = np.zeros(1, yt.camera_dt)
    for aspect, res, want in [(16 / 9, 1280, (1280, 720)), (1.0, 256, (256, 256)), (0.5, 256, (128, 256))]:
        cam["aspect"] = aspect
        assert yo.state_size(cam[0], res) == want


def test_hit_index_known_answer(bundles):
    """Cornell 256^2 sample-0 primary rays, FNV-1a over (instance,
    element) = e32777791f766a25, sum of distances 264850.978779."""
    b = bundles("cornellbox")
    p = yt.trace_params(resolution=256, sampler="falsecolor", samples=1)
    rays = b.camera_rays(p)
    hits = b.intersect_batch(rays)
    assert hits["hit"].sum() == 65536
    pairs = np.stack([hits["instance"], hits["element"]], 1).astype("<i4")
    assert ry.fnv1a64(pairs.tobytes()) == 0xE32777791F766A25 == int(KAT["hit_hash"])
    assert abs(hits["distance"].astype("f8").sum() - 264850.978779) < 1e-3


def test_image_hash_known_answer(bundles):
    """SURVEY.md Appendix A matrix: Cornell 128x128x8 spp `path`:
    FNV-1a(state.image) = a2d4acb1a35ee25c, FNV-1a(state.rngs) = 57e409c41ef42565."""
    b = bundles("cornellbox")
    p = yt.trace_params(resolution=128, sampler="path", samples=8, batch=8)
    st = b.render(p)
    assert ry.fnv1a64(st["image"].tobytes()) == 0xA2D4ACB1A35EE25C == int(KAT["image_hash"])
    assert ry.fnv1a64(st["rngs"].tobytes()) == 0x57E409C41EF42565 == int(KAT["rngs_hash"])


@pytest.mark.parametrize("name", list(P.SCENES))
def test_hits_match_golden(bundles, name):
    b = bundles(name)
    g = np.load(os.path.join(P.GOLDEN, f"hits_{name}.npz"))
    rays = P.random_rays(b.flat, 2048)
    assert b.intersect_batch(rays).tobytes() == g["hits"].tobytes()
    assert b.intersect_batch(rays, find_any=True).tobytes() == g["hits_any"].tobytes()
    assert b.intersect_instance_batch(g["inst"], rays).tobytes() == g["hits_inst"].tobytes()


@pytest.mark.parametrize("name,sampler", [(n, s) for n, ss in RENDERABLE.items() for s in ss])
def test_renders_match_golden(bundles, name, sampler):
    b = bundles(name)
    p = yt.trace_params(sampler=sampler, resolution=32, samples=4, batch=4, falsecolor="normal")
    st = b.render(p)
    g = np.load(os.path.join(P.GOLDEN, f"render_{name}_{sampler}.npz"))
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert st[k].tobytes() == g[k].tobytes(), k


def test_unrestated_scenes_are_refused_loudly(bundles):
    b = bundles("materials")  # glossy / refractive / textures ...
    p = yt.trace_params(sampler="path", resolution=16, samples=1)
    assert not b.supported(p)
    with pytest.raises(yo.OracleError):
        b.render(p)
    with pytest.raises(yo.OracleError):  # a sampler outside the restatement
        bundles("cornellbox").render(yt.trace_params(sampler="pathmis", resolution=16, samples=1))


@needs_ref
@pytest.mark.parametrize("name", list(P.SCENES))
def test_hits_match_live_reference(bundles, name):
    b = bundles(name)
    rb = P.RefBundle(b.flat)
    rays = P.random_rays(b.flat, 20000, seed=5)
    assert b.intersect_batch(rays).tobytes() == ry.intersect_batch(rb.bvh, rb.scene, rays).tobytes()
    inst = (np.arange(len(rays)) % len(b.flat.instances)).astype("i4")
    assert b.intersect_instance_batch(inst, rays).tobytes() == \
        ry.intersect_instance_batch(rb.bvh, rb.scene, inst, rays).tobytes()


@needs_ref
@pytest.mark.parametrize("name,sampler,kw", [
    ("cornellbox", "path", {}), ("cornellbox", "path", dict(tentfilter=1, nocaustics=1)),
    ("cornellbox", "naive", dict(envhidden=1)), ("cornellbox", "eyelight", {}),
    ("plane", "path", {}), ("plane", "naive", {}), ("plane", "path", dict(envhidden=1, bounces=3))])
def test_renders_match_live_reference(bundles, name, sampler, kw):
    """Whole trace_state, progressive (two batches), bit for bit."""
    b = bundles(name)
    p = yt.trace_params(sampler=sampler, resolution=48, samples=6, batch=3, **kw)
    st = b.render(p)
    ref = P.RefBundle(b.flat).render(p)
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert st[k].tobytes() == ref[k].tobytes(), k


@needs_ref
def test_camera_rays_match_live_reference(bundles):
    for name in ["cornellbox", "materials"]:  # materials: aperture > 0 (lens sampling)
        b = bundles(name)
        for tent in [0, 1]:
            p = yt.trace_params(resolution=40, samples=1, tentfilter=tent)
            rb = P.RefBundle(b.flat)
            st = ry.RefState(rb.scene, p)
            assert b.camera_rays(p).tobytes() == ry.camera_rays(st, rb.scene, p).tobytes()
