"""GPU parity tests (`-m gpu`): the HIP path, called through the C ABI, against
  (a) the compiled reference live (oracle/_ref, when it travelled with the repo), and
  (b) the committed golden fixtures generated from it (tests/golden/).

Bars:  bit-exact for integer/index work (PCG streams, hit records, BVH, hits
counters) and for every float stage that involves no libm transcendental
(camera rays with aperture 0, traversal uv/distance, eyelight images on
polygonal matte scenes); radiance of the Monte Carlo integrators within the
float tolerances written in each test (device libm sin/cos/pow/exp/log differ
from glibc in the last ulp, and one ulp in a bounce direction can flip a hit)."""
import os

import numpy as np
import pytest

import parity as P
from parity import ry, yt

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not P.have_ref(), reason="oracle/_ref did not travel")

ALL_SCENES = list(P.SCENES)


@pytest.fixture(scope="module")
def bundles():
    cache = {}

    def get(name):
        if name not in cache:
            flat = P.SCENES[name]()
            cache[name] = (flat, P.gpu_context(flat), P.RefBundle(flat) if P.have_ref() else None)
        return cache[name]

    yield get
    for _, ctx, _ in cache.values():
        ctx.close()


def test_native_library_is_loaded_and_no_fallback():
    lib = yt.load_library()
    assert lib._name.endswith("libythip.so")
    ctx = yt.Context(0)
    # trace before residency must fail loudly
    with pytest.raises(yt.YthipError):
        ctx.trace_samples(yt.trace_params())
    ctx.close()


# ---------------------------------------------------------------------------
# BVH residency
# ---------------------------------------------------------------------------
@needs_ref
@pytest.mark.parametrize("name", ALL_SCENES)
def test_resident_bvh_equals_reference_tree(bundles, name):
    flat, ctx, rb = bundles(name)
    assert ctx.download_bvh().same_as(rb.bvh.flat())
    lights = ctx.download_lights()
    rl = rb.lights.flat()
    assert lights.lights.tobytes() == rl.lights.tobytes() and lights.cdf.tobytes() == rl.cdf.tobytes()


@needs_ref
def test_uploaded_reference_bvh_gives_same_hits(bundles):
    """Drop-in path: trees built by the reference's make_trace_bvh and uploaded."""
    flat, ctx, rb = bundles("materials")
    ctx2 = yt.Context(0)
    ctx2.upload_scene(flat)
    ctx2.upload_bvh(rb.bvh.flat())
    rays = P.random_rays(flat, 4096, seed=5)
    assert P.hits_equal(ctx2.intersect_batch(rays), ctx.intersect_batch(rays))
    ctx2.close()


# ---------------------------------------------------------------------------
# traversal: bit-exact hit records
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ALL_SCENES)
def test_intersect_batch_golden(bundles, name):
    flat, ctx, _ = bundles(name)
    g = np.load(os.path.join(P.GOLDEN, f"hits_{name}.npz"))
    rays = P.random_rays(flat, 2048)
    assert P.hits_equal(ctx.intersect_batch(rays), g["hits"])
    assert P.hits_equal(ctx.intersect_batch(rays, find_any=True), g["hits_any"])
    assert P.hits_equal(ctx.intersect_instance_batch(g["inst"], rays), g["hits_inst"])
    assert g["hits"]["hit"].sum() > 50  # the batch is not vacuous


@needs_ref
@pytest.mark.parametrize("name", ALL_SCENES)
def test_intersect_batch_live_large(bundles, name):
    flat, ctx, rb = bundles(name)
    rays = P.random_rays(flat, 100_000, seed=17)
    a = ctx.intersect_batch(rays)
    b = ry.intersect_batch(rb.bvh, rb.scene, rays)
    assert P.hits_equal(a, b)
    inst = (np.arange(len(rays)) * 7 % len(flat.instances)).astype("i4")
    assert P.hits_equal(ctx.intersect_instance_batch(inst, rays),
                        ry.intersect_instance_batch(rb.bvh, rb.scene, inst, rays))
    assert P.hits_equal(ctx.intersect_batch(rays, find_any=True),
                        ry.intersect_batch(rb.bvh, rb.scene, rays, find_any=True))


def test_cornell_hit_index_known_answer():
    """Cornell 256², sample-0 primary rays: FNV-1a-64 over
    (instance, element) = e32777791f766a25, Σdistance = 264850.978779."""
    flat = P.SCENES["cornellbox"]()
    ctx = P.gpu_context(flat)
    p = yt.trace_params(resolution=256, samples=1)
    ctx.make_trace_state(flat, p)
    rays = ctx.camera_rays(p)
    hits = ctx.intersect_batch(rays)
    pairs = np.stack([hits["instance"], hits["element"]], 1).astype("<i4")
    assert hits["hit"].sum() == 65536
    assert ry.fnv1a64(pairs.tobytes()) == 0xE32777791F766A25
    assert abs(hits["distance"].astype("f8").sum() - 264850.978779) < 1e-3
    ctx.close()


def test_empty_and_degenerate_inputs():
    flat = P.SCENES["cornellbox"]()
    ctx = P.gpu_context(flat)
    assert len(ctx.intersect_batch(np.zeros(0, yt.ray_dt))) == 0
    rays = np.zeros(4, yt.ray_dt)  # zero direction: 1/0 = inf, NaNs in the slab test
    rays["tmin"], rays["tmax"] = 1e-4, np.finfo("f4").max
    rays["o"] = [0, 1, 0]
    h = ctx.intersect_batch(rays)
    if P.have_ref():
        rb = P.RefBundle(flat)
        assert P.hits_equal(h, ry.intersect_batch(rb.bvh, rb.scene, rays))
    # a scene whose only instance has no elements, and an empty scene
    sc = yt.FlatScene()
    sc.add_camera(yt.IDENTITY_FRAME)
    sc.add_material("matte", color=(1, 1, 1))
    s = sc.add_shape(np.zeros((3, 3), "f4"))
    sc.add_instance(s, 0)
    c2 = P.gpu_context(sc)
    r = P.random_rays(flat, 64)
    assert c2.intersect_batch(r)["hit"].sum() == 0
    c2.close()
    ctx.close()


# ---------------------------------------------------------------------------
# camera + PCG: bit-exact
# ---------------------------------------------------------------------------
@needs_ref
@pytest.mark.parametrize("name,tent", [("cornellbox", False), ("cornellbox", True),
                                       ("plane", False), ("lines_points", True)])
def test_camera_rays_bit_exact(bundles, name, tent):
    flat, ctx, rb = bundles(name)
    p = yt.trace_params(resolution=96, samples=1, tentfilter=int(tent))
    ctx.make_trace_state(flat, p)
    st = ry.RefState(rb.scene, p)
    a = ctx.camera_rays(p)
    b = ry.camera_rays(st, rb.scene, p)
    assert a.tobytes() == b.tobytes()


# ---------------------------------------------------------------------------
# images
# ---------------------------------------------------------------------------
def _render_gpu(ctx, flat, **kw):
    params = yt.trace_params(**kw)
    return P.gpu_render(ctx, flat, params), params


@pytest.mark.parametrize("name", ["cornellbox", "plane", "instances"])
def test_eyelight_image_bit_exact_golden(bundles, name):
    """eyelight on polygonal matte/glossy-free scenes involves no libm call →
    the whole trace_state (image, albedo, normal, hits, rngs) must be bit-identical."""
    flat, ctx, _ = bundles(name)
    g = np.load(os.path.join(P.GOLDEN, f"render_{name}_eyelight.npz"))
    st, _ = _render_gpu(ctx, flat, sampler="eyelight", resolution=32, samples=4, batch=4,
                        falsecolor="normal")
    assert st["rngs"].tobytes() == g["rngs"].tobytes()
    assert np.array_equal(st["hits"], g["hits"])
    if name == "instances":  # glossy instances: fresnel/microfacet are libm-free too
        pass
    assert st["image"].tobytes() == g["image"].tobytes()
    assert st["albedo"].tobytes() == g["albedo"].tobytes()
    assert st["normal"].tobytes() == g["normal"].tobytes()


@pytest.mark.parametrize("name,sampler", [(n, s) for n, ss in {
    "cornellbox": ["path", "pathdirect", "pathmis", "pathtest", "naive", "eyelight",
                   "diagram", "furnace", "falsecolor"],
    "materials": ["path", "pathdirect", "pathmis", "naive", "eyelight", "falsecolor"],
    "plane": ["path", "eyelight"],
    "instances": ["path", "eyelight", "falsecolor"],
    "lines_points": ["path", "eyelight", "falsecolor"]}.items() for s in ss])
def test_render_vs_golden(bundles, name, sampler):
    """Same scene, seed, spp as the fixture (a render of the compiled reference): the whole
    trace_state bit for bit."""
    flat, ctx, _ = bundles(name)
    g = np.load(os.path.join(P.GOLDEN, f"render_{name}_{sampler}.npz"))
    st, _ = _render_gpu(ctx, flat, sampler=sampler, resolution=32, samples=4, batch=4,
                        falsecolor="normal")
    s = P.image_stats(st["image"], g["image"])
    same_rng = float((st["rngs"] == g["rngs"]).all(axis=1).mean())
    assert np.isfinite(st["image"]).all()
    assert same_rng == 1.0, (same_rng, s)
    P.assert_identical(st, g, f"{name} {sampler} vs golden")


@needs_ref
@pytest.mark.parametrize("name,sampler,res,spp", [
    ("cornellbox", "path", 128, 16), ("cornellbox", "pathmis", 64, 8),
    ("materials", "path", 96, 16), ("materials", "pathdirect", 64, 8),
    ("plane", "path", 128, 8), ("instances", "path", 96, 8),
    ("lines_points", "path", 96, 8), ("materials", "naive", 64, 8),
    ("materials", "furnace", 64, 4), ("materials", "pathtest", 64, 4),
])
def test_render_vs_live_reference(bundles, name, sampler, res, spp):
    """Larger live comparison against the compiled reference, progressive
    (batch < samples, so state carries across trace_samples calls)."""
    flat, ctx, rb = bundles(name)
    params = yt.trace_params(sampler=sampler, resolution=res, samples=spp, batch=max(spp // 4, 1))
    gpu = P.gpu_render(ctx, flat, params)
    ref = rb.render(params)
    s = P.image_stats(gpu["image"], ref["image"])
    same_rng = float((gpu["rngs"] == ref["rngs"]).all(axis=1).mean())
    assert gpu["samples"] == ref["samples"] == spp
    assert same_rng == 1.0, (same_rng, s)
    P.assert_identical(gpu, ref, f"{name} {sampler}")


@needs_ref
@pytest.mark.parametrize("fc", ["position", "normal", "frontfacing", "gnormal", "gfrontfacing",
                                "texcoord", "color", "emission", "roughness", "opacity",
                                "metallic", "delta", "mtype", "instance", "shape", "material",
                                "element", "highlight"])
def test_falsecolor_modes(bundles, fc):
    """Every falsecolor mode (yocto_trace.cpp:1366-1415), bit for bit (the only libm call is
    the final srgb_to_rgb's powf — glibc's, restated in yt_libm.h)."""
    flat, ctx, rb = bundles("materials")
    params = yt.trace_params(sampler="falsecolor", falsecolor=fc, resolution=64, samples=1)
    gpu = P.gpu_render(ctx, flat, params)
    ref = rb.render(params)
    assert gpu["rngs"].tobytes() == ref["rngs"].tobytes()
    assert np.array_equal(gpu["hits"], ref["hits"])
    P.assert_identical(gpu, ref, f"falsecolor {fc}")


@needs_ref
def test_params_variants(bundles):
    """envhidden / nocaustics / clamp / bounces / seed / second camera-less edge."""
    flat, ctx, rb = bundles("materials")
    for kw in [dict(envhidden=1), dict(nocaustics=1), dict(clamp=0.5), dict(bounces=2),
               dict(bounces=1), dict(seed=12345), dict(tentfilter=1)]:
        params = yt.trace_params(sampler="path", resolution=48, samples=4, batch=2, **kw)
        gpu = P.gpu_render(ctx, flat, params)
        ref = rb.render(params)
        s = P.image_stats(gpu["image"], ref["image"])
        same_rng = float((gpu["rngs"] == ref["rngs"]).all(axis=1).mean())
        assert same_rng == 1.0, (kw, same_rng, s)
        P.assert_identical(gpu, ref, str(kw))


@needs_ref
def test_resume_cpu_state_on_gpu_and_back(bundles):
    """trace_state is the checkpoint: a render started on the CPU
    reference continues on the GPU and vice versa (eyelight → bit-exact)."""
    flat, ctx, rb = bundles("cornellbox")
    params = yt.trace_params(sampler="eyelight", resolution=48, samples=4, batch=2)
    st = ry.RefState(rb.scene, params)
    ry.trace_samples(st, rb.scene, rb.bvh, rb.lights, params)  # samples 0-1 on the CPU
    half = st.get()
    ctx.make_trace_state(flat, params)
    ctx.upload_state(half["image"], half["albedo"], half["normal"], half["hits"], half["rngs"],
                     samples=half["samples"])
    ctx.trace_samples(params)  # samples 2-3 on the GPU
    ctx.trace_samples(params)  # no-op: samples >= params.samples (yocto_trace.cpp:1598)
    gpu = ctx.download_state()
    ry.trace_samples(st, rb.scene, rb.bvh, rb.lights, params)
    ref = st.get()
    assert gpu["samples"] == 4
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert gpu[k].tobytes() == ref[k].tobytes(), k


def test_row_sharding_equals_full_frame(bundles):
    """§8e: rendering rows [a,b) with the sliced seeds gives exactly the rows of
    the full-frame render (no cross-pixel data flow)."""
    flat, ctx, _ = bundles("materials")
    params = yt.trace_params(sampler="path", resolution=64, samples=2, batch=2)
    full = P.gpu_render(ctx, flat, params)
    w, h = full["width"], full["height"]
    rngs = yt.make_rngs(params.seed, w * h)
    parts = []
    for r0, r1 in [(0, 5), (5, h // 2), (h // 2, h)]:
        parts.append(P.gpu_render(ctx, flat, params, rows=(r0, r1), rngs=rngs))
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert np.concatenate([p[k] for p in parts]).tobytes() == full[k].tobytes(), k


def test_column_striped_sharding_equals_full_frame(bundles):
    """§8e, balanced slicing: rank r of G renders tile columns r, r+G, ... (and a
    row range); put back through the shard's pixel list the slices tile the
    full-frame render bit for bit, for every state array."""
    import sharding
    flat, ctx, _ = bundles("materials")
    params = yt.trace_params(sampler="path", resolution=72, samples=2, batch=2)
    full = P.gpu_render(ctx, flat, params)
    w, h = full["width"], full["height"]
    rngs = yt.make_rngs(params.seed, w * h)
    for world in [2, 3]:
        seen = np.zeros(w * h, bool)
        for r in range(world):
            sh = sharding.shard_frame(w, h, world, r, "columns")
            part = P.gpu_render(ctx, flat, params, rows=sh.rows, cols=sh.cols, rngs=rngs)
            assert part["width"] == sh.local_width
            for k in ["image", "albedo", "normal", "hits", "rngs"]:
                assert part[k].tobytes() == full[k][sh.pixels].tobytes(), (k, world, r)
            seen[sh.pixels] = True
        assert seen.all()
    # rows and columns together, camera rays included
    sh = sharding.Shard(w, h, (7, 30), (1, 2))
    part = P.gpu_render(ctx, flat, params, rows=sh.rows, cols=sh.cols, rngs=rngs)
    assert part["image"].tobytes() == full["image"][sh.pixels].tobytes()
    ctx.make_trace_state(flat, params, rngs=rngs)
    rays_full = ctx.camera_rays(params)
    ctx.make_trace_state(flat, params, rows=sh.rows, cols=sh.cols, rngs=rngs)
    assert ctx.camera_rays(params).tobytes() == rays_full[sh.pixels].tobytes()


@pytest.mark.parametrize("name", ALL_SCENES)
def test_hits_bit_exact_vs_cpu_restatement(bundles, name):
    """HIP traversal vs the CPU restatement of intersect_scene_bvh /
    intersect_instance_bvh (oracle/yt_oracle.cpp) on 100k seeded rays — needs
    nothing from /root/reference on the GPU box."""
    import ytoracle as yo
    flat, ctx, _ = bundles(name)
    ob = yo.Bundle(flat)
    rays = P.random_rays(flat, 100000, seed=17)
    assert ctx.intersect_batch(rays).tobytes() == ob.intersect_batch(rays).tobytes()
    assert ctx.intersect_batch(rays, find_any=True).tobytes() == ob.intersect_batch(rays, find_any=True).tobytes()
    inst = (np.arange(len(rays)) % len(flat.instances)).astype("i4")
    assert ctx.intersect_instance_batch(inst, rays).tobytes() == ob.intersect_instance_batch(inst, rays).tobytes()


@pytest.mark.parametrize("name", ["cornellbox", "plane"])
def test_renders_vs_cpu_restatement(bundles, name):
    """Whole trace_state vs the CPU restatement: eyelight bit-exact (no libm on
    the path); path within the stated tolerance — >= 97 % of the pixels keep the
    oracle's rng stream and agree to 1e-4 relative, image means within 1 %."""
    import ytoracle as yo
    flat, ctx, _ = bundles(name)
    ob = yo.Bundle(flat)
    p = yt.trace_params(sampler="eyelight", resolution=96, samples=3, batch=3)
    gpu, cpu = P.gpu_render(ctx, flat, p), ob.render(p)
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert gpu[k].tobytes() == cpu[k].tobytes(), k
    p = yt.trace_params(sampler="path", resolution=96, samples=4, batch=2)
    gpu, cpu = P.gpu_render(ctx, flat, p), ob.render(p)
    P.assert_identical(gpu, cpu, f"{name} path vs the CPU restatement")


@pytest.mark.parametrize("name,sampler", [("plane", "path"), ("plane", "naive"), ("cornellbox", "path"),
                                          ("materials", "pathmis"), ("lines_points", "path")])
def test_results_do_not_depend_on_the_scheduling_policy(bundles, name, sampler):
    """k_trace's adaptive wait (camera rays wait for cheap bounce rays)
    only reorders independent pixels' work: every state array is bit-identical
    with the policy on and off."""
    flat, ctx, _ = bundles(name)
    p = yt.trace_params(sampler=sampler, resolution=160, samples=6, batch=3)
    out = []
    for wait in [1, 0]:
        ctx.set_scheduling(wait)
        out.append(P.gpu_render(ctx, flat, p))
    ctx.set_scheduling(1)
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert out[0][k].tobytes() == out[1][k].tobytes(), k


@pytest.mark.parametrize("name,sampler,kw", [
    ("plane", "path", {}), ("plane", "pathtest", {}), ("plane", "path", {"envhidden": 1}),
    ("plane", "path", {"bounces": 1}), ("plane", "path", {"bounces": 2, "tentfilter": 1}),
    ("plane", "naive", {}), ("plane", "naive", {"envhidden": 1}), ("plane", "eyelight", {}),
    ("materials", "naive", {}), ("materials", "eyelight", {"envhidden": 1}),
    ("cornellbox", "path", {}), ("materials", "path", {}), ("instances", "path", {}),
    ("lines_points", "path", {})])
def test_results_do_not_depend_on_the_early_miss(bundles, name, sampler, kw):
    """resolve_step's early miss (a continuing path whose next ray cannot enter the
    scene's root box takes its miss branch in place) performs the same
    operations per path in the same order: every state array is bit-identical with it
    on and off — open scenes where it fires for most bounce rays (the plane), with
    envhidden / the bounce limit deciding whether the environment is added, opacity
    retries and volumes (materials), closed scenes where it never fires."""
    flat, ctx, _ = bundles(name)
    p = yt.trace_params(sampler=sampler, resolution=160, samples=6, batch=3, **kw)
    out = []
    for on in [1, 0]:
        ctx.set_early_miss(on)
        out.append(P.gpu_render(ctx, flat, p))
    ctx.set_early_miss(1)
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert out[0][k].tobytes() == out[1][k].tobytes(), k


def test_early_miss_does_not_change_the_work_counts(bundles):
    """The counting launch keeps the plain flow: rays / nodes per sample are the
    reference's whether the early miss is on or off."""
    flat, ctx, _ = bundles("plane")
    p = yt.trace_params(sampler="path", resolution=96, samples=2, batch=2)
    stats = []
    for on in [1, 0]:
        ctx.set_early_miss(on)
        ctx.set_profiling(2)
        ctx.reset_stats()
        P.gpu_render(ctx, flat, p)
        stats.append(ctx.get_stats())
        ctx.set_profiling(0)
    ctx.set_early_miss(1)
    for k in ["samples", "rays", "nodes", "triangles", "instances", "shades"]:
        assert stats[0][k] == stats[1][k], k


@pytest.mark.parametrize("name", ALL_SCENES)
def test_wide_and_binary_walks_agree(bundles, name):
    """The wide walk (four grandchildren per fetch) and the binary walk give the same
    hit records on 100k seeded rays incl. degenerate directions (which the wide walk
    hands over to the binary one), single-instance walks, and the same images."""
    flat, ctx, _ = bundles(name)
    rays = P.random_rays(flat, 100000, seed=29)
    inst = (np.arange(len(rays)) % len(flat.instances)).astype("i4")
    p = yt.trace_params(sampler="path", resolution=96, samples=3, batch=3)
    out = {}
    for mode in ["binary", "wide"]:
        ctx.set_traversal(mode)
        out[mode] = (ctx.intersect_batch(rays), ctx.intersect_instance_batch(inst, rays), P.gpu_render(ctx, flat, p))
    ctx.set_traversal("auto")
    assert out["binary"][0].tobytes() == out["wide"][0].tobytes()
    assert out["binary"][1].tobytes() == out["wide"][1].tobytes()
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert out["binary"][2][k].tobytes() == out["wide"][2][k].tobytes(), k


@pytest.mark.parametrize("name", ["cornellbox", "plane"])
def test_matte_specialised_kernel_gives_identical_results(bundles, name):
    """All-matte scenes run a k_trace variant compiled without the other lobes: every
    state array is bit-identical to the general kernel's (area lights: the deferred
    light-pdf variant on the Cornell box; environment light on the plane)."""
    flat, ctx, _ = bundles(name)
    assert (flat.materials["type"] == 0).all()
    p = yt.trace_params(sampler="path", resolution=128, samples=6, batch=3)
    out = []
    for spec in [1, 0]:
        ctx.set_specialization(spec)
        out.append(P.gpu_render(ctx, flat, p))
    ctx.set_specialization(1)
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert out[0][k].tobytes() == out[1][k].tobytes(), k


@needs_ref
def test_8k_frame_bit_exact():
    """A large frame: 7680 x 4320 (33 M pixels, 2 GB of trace_state, 518,400 one-wave
    workgroups, int32 pixel indexing x 3 for the vec3f arrays) — eyelight, 1 spp, every
    state array bit for bit vs the reference."""
    flat = P.SCENES["plane"]()
    ctx, rb = P.gpu_context(flat), P.RefBundle(flat)
    p = yt.trace_params(sampler="eyelight", resolution=7680, samples=1, batch=1)
    gpu = P.gpu_render(ctx, flat, p)
    ctx.close()
    ref = rb.render(p)
    assert (gpu["width"], gpu["height"]) == (ref["width"], ref["height"]) == (7680, 4320)
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert gpu[k].tobytes() == ref[k].tobytes(), k
    assert int(gpu["hits"].sum()) == 7680 * 4320  # plane or environment: every sample counts


@needs_ref
@pytest.mark.parametrize("resolution,aspect", [(1, 1.0), (3, 1.0), (17, 0.5), (37, 1.7), (63, 1.0), (65, 2.9)])
def test_odd_frame_sizes_bit_exact(resolution, aspect):
    """Frames that do not fill the 16x4 tiles (1 pixel, 17x34, 37x22, 65x22 ...):
    eyelight trace_state bit for bit vs the reference, in two batches."""
    flat = P.SCENES["cornellbox"]()
    flat.cameras["aspect"] = aspect
    ctx, rb = P.gpu_context(flat), P.RefBundle(flat)
    p = yt.trace_params(sampler="eyelight", resolution=resolution, samples=4, batch=2)
    gpu, ref = P.gpu_render(ctx, flat, p), rb.render(p)
    ctx.close()
    assert (gpu["width"], gpu["height"]) == (ref["width"], ref["height"])
    for k in ["image", "albedo", "normal", "hits", "rngs"]:
        assert gpu[k].tobytes() == ref[k].tobytes(), (k, gpu["width"], gpu["height"])


def test_get_image_and_device_tonemap(bundles):
    """§8(f) rank 2, the display path: ythip_get_image returns exactly
    trace_state.image; ythip_tonemap_image (device) vs the reference's tonemap_image
    on the same pixels — floats and bytes identical for every curve (exp2f / powf are
    glibc's, csrc/yt_libm.h)."""
    flat, ctx, _ = bundles("materials")
    p = yt.trace_params(sampler="path", resolution=128, samples=4, batch=4)
    st = P.gpu_render(ctx, flat, p)
    img = ctx.get_image()
    assert img.reshape(-1, 4).tobytes() == st["image"].tobytes()
    ldr, ldrb = ctx.tonemap_image(exposure=0.0, filmic=False, srgb=False)
    assert ldr.tobytes() == img.tobytes()  # identity curve: bit-exact
    assert np.array_equal(ldrb.reshape(-1, 4), np.clip((img.reshape(-1, 4) * 256).astype(np.int64), 0, 255))
    if not P.have_ref():
        pytest.skip("oracle/_ref did not travel")
    for exposure, filmic in [(0.0, False), (1.5, False), (-0.75, True)]:
        ldr, ldrb = ctx.tonemap_image(exposure=exposure, filmic=filmic, srgb=True)
        ref, refb = ry.tonemap(img, exposure, filmic, True)
        assert ldr.reshape(-1, 4).tobytes() == ref.astype(np.float32).tobytes(), (exposure, filmic)  # powf / exp2f: glibc's
        assert np.array_equal(ldrb.reshape(-1, 4), refb), (exposure, filmic)


def test_guide_images_for_the_denoiser(bundles):
    """§8(f) rank 3, the denoiser hand-off: get_albedo_image / get_normal_image
    (yocto_trace.cpp:1769-1791) expanded on the device = {guide.xyz, 1} of the state,
    and equal to the reference's functions on the reference's own state (eyelight:
    the whole state is bit-exact)."""
    flat, ctx, rb = bundles("cornellbox")
    p = yt.trace_params(sampler="eyelight", resolution=96, samples=3, batch=3)
    st = P.gpu_render(ctx, flat, p)
    for get, key in [(ctx.get_albedo_image, "albedo"), (ctx.get_normal_image, "normal")]:
        img = get().reshape(-1, 4)
        assert img[:, :3].tobytes() == st[key].reshape(-1, 3).tobytes()
        assert (img[:, 3] == 1.0).all()
    if rb is None:
        pytest.skip("oracle/_ref did not travel")
    rst = ry.RefState(rb.scene, p)
    ry.trace_samples(rst, rb.scene, rb.bvh, rb.lights, p)
    assert ctx.get_albedo_image().tobytes() == ry.guide_image(rst, 0).tobytes()
    assert ctx.get_normal_image().tobytes() == ry.guide_image(rst, 1).tobytes()


@pytest.mark.parametrize("sampler", ["eyelight", "path", "falsecolor", "naive", "pathdirect", "pathmis"])
def test_trace_sample_single_pixels(bundles, sampler):
    """trace_sample (yocto_trace.cpp:1461-1492) through ythip_trace_sample: single
    pixels in arbitrary order (corners, ragged last tile, the same pixel twice) on top
    of a rendered state.  Untouched pixels keep their bytes; touched pixels equal the
    reference's trace_sample (bit for bit for eyelight / falsecolor and for the rng
    streams; path radiance within 1e-4 relative — device libm)."""
    flat, ctx, rb = bundles("cornellbox")
    p = yt.trace_params(sampler=sampler, resolution=70, samples=2, batch=2)
    before = P.gpu_render(ctx, flat, p)
    w, h = before["width"], before["height"]
    pixels = [(0, 0), (w - 1, h - 1), (17, 3), (64, h - 2), (35, 35), (35, 35), (16, 4)]
    for k, (i, j) in enumerate(pixels):
        ctx.trace_sample(p, i, j, 2 + k)
    after = ctx.download_state()
    assert after["samples"] == 2  # trace_sample leaves state.samples alone
    touched = np.zeros(w * h, bool)
    touched[[j * w + i for i, j in pixels]] = True
    for key in ["image", "albedo", "normal", "hits", "rngs"]:
        a, b = after[key].reshape(w * h, -1), before[key].reshape(w * h, -1)
        assert a[~touched].tobytes() == b[~touched].tobytes(), key
    assert (after["rngs"].reshape(w * h, -1)[touched] != before["rngs"].reshape(w * h, -1)[touched]).any(1).all()
    with pytest.raises(yt.YthipError):
        ctx.trace_sample(p, w, 0, 0)
    with pytest.raises(yt.YthipError):
        ctx.trace_sample(p, 0, 0, -1)
    if rb is None:
        pytest.skip("oracle/_ref did not travel")
    rst = ry.RefState(rb.scene, p)
    rst.set(**{k: before[k] for k in ["image", "albedo", "normal", "hits", "rngs"]}, samples=2)
    for k, (i, j) in enumerate(pixels):
        ry.trace_sample(rst, rb.scene, rb.bvh, rb.lights, p, i, j, 2 + k)
    ref = rst.get()
    assert after["rngs"].tobytes() == ref["rngs"].tobytes()
    assert after["hits"].tobytes() == ref["hits"].tobytes()
    for key in ["image", "albedo", "normal"]:
        assert after[key].tobytes() == ref[key].tobytes(), key


def test_trace_sample_on_a_column_striped_slice(bundles):
    """The same through a slice of the frame (rank 1 of 3, tile columns 1, 4, ...): a pixel
    of the slice equals the full-frame result, a pixel of another rank is refused."""
    flat, ctx, _ = bundles("cornellbox")
    p = yt.trace_params(sampler="eyelight", resolution=100, samples=1, batch=1)
    full = P.gpu_render(ctx, flat, p)
    w, h = full["width"], full["height"]
    i, j = 4 * 16 + 5, 41  # tile column 4 = local tile column 1 of rank 1/3
    ctx.trace_sample(p, i, j, 1)
    want = ctx.download_state()
    rngs = yt.make_rngs(p.seed, w * h)
    sl = P.gpu_render(ctx, flat, p, rngs=rngs, cols=(1, 3))
    lw = sl["width"]
    ctx.trace_sample(p, i, j, 1)
    got = ctx.download_state()
    il = 1 * 16 + 5
    for key in ["image", "albedo", "normal", "hits", "rngs"]:
        a = got[key].reshape(h * lw, -1)[j * lw + il]
        b = want[key].reshape(h * w, -1)[j * w + i]
        assert a.tobytes() == b.tobytes(), key
    with pytest.raises(yt.YthipError):
        ctx.trace_sample(p, 5, j, 1)  # tile column 0 belongs to rank 0


def test_work_counters_and_cancel(bundles):
    flat, ctx, _ = bundles("cornellbox")
    params = yt.trace_params(sampler="path", resolution=64, samples=2, batch=2)
    ctx.set_profiling(3)
    ctx.reset_stats()
    P.gpu_render(ctx, flat, params)
    s = ctx.get_stats()
    ctx.set_profiling(0)
    assert s["samples"] == 64 * 64 * 2 and s["rays"] >= s["samples"]
    assert s["nodes"] > s["rays"] and s["triangles"] > 0 and s["instances"] > 0
    assert s["trace_launches"] > 0 and s["trace_ms"] > 0
    assert yt.algorithmic_bytes(s) > 0
    # cancellation (trace_start's stop flag, yocto_trace.cpp:1637)
    ctx.make_trace_state(flat, params)
    stop = np.ones(1, "i4")
    with pytest.raises(yt.YthipError):
        ctx.trace_samples(params, stop=stop)
    with pytest.raises(yt.YthipError):
        ctx.trace_samples(yt.trace_params(sampler=42))  # "sampler unknown"


# ---------------------------------------------------------------------------
# the C++ drop-in boundary (yocto-gl_amd/host/yocto_hiptrace.h)
# ---------------------------------------------------------------------------
DROPIN = os.path.join(os.path.dirname(P.GOLDEN.rstrip("/")), "..", "oracle", "_ref", "dropin_test")


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/dropin_test did not travel")
def test_cpp_dropin_matches_reference_api():
    """The reference's own scene_data / trace_bvh / trace_lights / trace_state objects
    through yocto::hip::trace_samples vs yocto::trace_samples, in C++
    (tests/cpp/dropin_test.cpp): eyelight bit-exact incl. progressive batches,
    CPU<->GPU resume, camera edits, trace_image, exception behaviour; path within
    the tolerances stated there."""
    import subprocess
    r = subprocess.run([DROPIN], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dropin_test: OK" in r.stdout and "devices: 1" in r.stdout


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/dropin_test did not travel")
def test_cpp_dropin_sharded_over_two_ranks():
    """The same C++ program with YOCTO_HIP_DEVICES=0,0: the shim drives TWO ranks through
    libythip's ythip_multi (tile columns dealt round-robin, per-rank contexts and streams,
    concurrent launches, the framebuffer gather + un-permute) — here on one physical GPU,
    where the gather uses device copies (RCCL refuses a communicator with a duplicate
    device).  Every comparison with the CPU reference must hold exactly as with one rank."""
    import subprocess
    env = dict(os.environ, YOCTO_HIP_DEVICES="0,0")
    r = subprocess.run([DROPIN], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dropin_test: OK" in r.stdout and "devices: 2" in r.stdout


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/dropin_test did not travel")
@pytest.mark.parametrize("devices", ["0", "0,0"])
def test_cpp_dropin_on_the_streaming_scheduler(devices):
    """... and with the streaming scheduler as the library's default (YTHIP_SCHEDULER=1; the wide walk forced so that the small test
    scenes are served): `path` batches of 4 and of 256 samples and a cancelled render run streamed, on one rank and on two — where
    ythip_multi launches the ranks from a thread each, since a streamed batch returns when it is done.  Same comparisons."""
    import subprocess
    env = dict(os.environ, YOCTO_HIP_DEVICES=devices, YTHIP_SCHEDULER="1", YTHIP_TRAVERSAL="1", YTHIP_STREAM_MIN_SLOTS="1024")
    r = subprocess.run([DROPIN], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dropin_test: OK" in r.stdout


YTRACE_CPU = os.path.join(os.path.dirname(DROPIN), "ytrace_cpu")
YTRACE_HIP = os.path.join(os.path.dirname(DROPIN), "ytrace_hip")


@pytest.mark.skipif(not (os.path.exists(YTRACE_CPU) and os.path.exists(YTRACE_HIP) and P.have_ref()),
                    reason="oracle/_ref/ytrace_{cpu,hip} did not travel")
@pytest.mark.parametrize("scene,args,exact", [
    ("cornellbox", ["--sampler", "eyelight", "--samples", "4", "--resolution", "160"], True),
    ("cornellbox", ["--sampler", "falsecolor", "--falsecolor", "normal", "--samples", "2", "--batch", "2",
                    "--resolution", "96"], True),
    ("cornellbox", ["--sampler", "path", "--samples", "64", "--batch", "16", "--resolution", "128"], False),
    ("materials", ["--sampler", "path", "--samples", "32", "--batch", "8", "--resolution", "128",
                   "--tentfilter"], False),
    ("instances", ["--sampler", "eyelight", "--samples", "3", "--resolution", "200"], True),
    ("lines_points", ["--sampler", "path", "--samples", "32", "--batch", "32", "--resolution", "128"], False)])
def test_unmodified_ytrace_app_on_both_backends(tmp_path, scene, args, exact):
    """The reference's own command-line renderer, apps/ytrace.cpp, compiled UNMODIFIED
    twice (oracle/Makefile): as it is (ytrace_cpu) and with the force-included prelude
    that redirects its six trace_* calls to yocto::hip (ytrace_hip,
    yocto-gl_amd/host/ytrace_hip_prelude.h).  Both load the same scene file — written
    by the reference's save_scene — and save an .hdr that the reference's load_image
    reads back: identical pixels, whatever the sampler (the device evaluates the
    reference platform's libm, csrc/yt_libm.h)."""
    import subprocess
    sc = ry.RefScene.from_flat(P.SCENES[scene]())
    fn = tmp_path / scene / (scene + ".json")
    os.makedirs(fn.parent, exist_ok=True)
    sc.save(fn)
    out = {}
    for exe in [YTRACE_CPU, YTRACE_HIP]:
        o = tmp_path / (os.path.basename(exe) + ".hdr")
        r = subprocess.run([exe, "--scene", str(fn), "--output", str(o)] + args, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "render image" in r.stdout
        out[exe] = ry.load_image(o)
    a, b = out[YTRACE_CPU], out[YTRACE_HIP]
    assert a.shape == b.shape and a.shape[0] > 0
    assert a.tobytes() == b.tobytes()  # the saved images are the same files, whatever the sampler


@pytest.mark.skipif(not (os.path.exists(YTRACE_CPU) and os.path.exists(YTRACE_HIP) and P.have_ref()),
                    reason="oracle/_ref/ytrace_{cpu,hip} did not travel")
@pytest.mark.parametrize("scheduler", ["1", "2"])
def test_unmodified_ytrace_app_on_the_streaming_scheduler(tmp_path, scheduler):
    """The drop-in takes the library's defaults from the environment: YTHIP_SCHEDULER=1 (streamed wherever served) / 2 (the
    measured choice: fused, fused timed, streamed, streamed timed, then the faster) under the unmodified ytrace — the same
    .hdr as the CPU build's, byte for byte.  (YTHIP_TRAVERSAL=1: the test scene's trees are tiny and would be walked binary,
    which the scheduler does not serve.)"""
    import subprocess
    sc = ry.RefScene.from_flat(P.SCENES["cornellbox"]())
    fn = tmp_path / "cornellbox" / "cornellbox.json"
    os.makedirs(fn.parent, exist_ok=True)
    sc.save(fn)
    args = ["--sampler", "path", "--samples", "96", "--batch", "8", "--resolution", "128"]
    out = {}
    for exe in [YTRACE_CPU, YTRACE_HIP]:
        o = tmp_path / (os.path.basename(exe) + ".hdr")
        env = dict(os.environ, YTHIP_SCHEDULER=scheduler, YTHIP_TRAVERSAL="1", YTHIP_STREAM_MIN_SLOTS="1024")
        r = subprocess.run([exe, "--scene", str(fn), "--output", str(o)] + args, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        out[exe] = ry.load_image(o)
    assert out[YTRACE_CPU].tobytes() == out[YTRACE_HIP].tobytes()


def test_no_device_memory_leak_over_context_and_state_cycles():
    """Contexts, scenes, trees, lights and states come and go (an interactive session
    rebuilds the state on every camera edit, apps/ytrace.cpp:189-204): device memory
    returns to where it was."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")  # already in the process (libythip links it)

    def free_bytes():
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    flat = P.SCENES["plane"]()

    def cycle(n_states):
        ctx = P.gpu_context(flat)
        for k in range(n_states):
            q = yt.trace_params(sampler="pathmis" if k % 2 else "path", resolution=128 + 16 * (k % 5),
                                samples=2, batch=2)
            P.gpu_render(ctx, flat, q)
        ctx.tonemap_image()
        ctx.get_albedo_image()
        ctx.close()

    cycle(2)  # warm-up: runtime pools, code objects
    free0 = free_bytes()
    for _ in range(10):
        cycle(12)
    free1 = free_bytes()
    assert free0 - free1 < 64 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB"
