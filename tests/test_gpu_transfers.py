"""Host <-> device transfers (`-m gpu`).  Round 2 lost three first GPU processes of fresh boxes to
`Memory access fault by GPU ... on address <host address>`: large pageable copies are not staged
by the HIP runtime, it pins the CALLER's pages on the fly (userptr) and DMAs from them in place.
Since round 3 every byte moves through pinned memory the library owns (csrc/yt_xfer.h).  What that
promises is tested here: the caller's buffer is free the moment an upload call returns, chunked
round trips of every size class are exact, pinned caller memory is still taken as it is, and a
fresh process that starts with the large uploads survives — over and over."""
import gc
import os
import subprocess
import sys

import numpy as np
import pytest

import parity as P
import ythip as yt

pytestmark = pytest.mark.gpu
ROOT = P.ROOT


def test_upload_does_not_depend_on_the_callers_buffers_after_the_call():
    """upload_scene / make_trace_bvh from pageable arrays that are scribbled over and freed right
    after each call: the render equals the one from untouched arrays."""
    import scenes as ysc
    params = yt.trace_params(sampler="path", resolution=320, samples=4, batch=4)
    flat = ysc.plane_scene()
    ctx = P.gpu_context(flat)
    want = P.gpu_render(ctx, flat, params)
    ctx.close()

    victim = ysc.plane_scene()
    ctx = yt.Context(0)
    ctx.upload_scene(victim)
    ctx.make_trace_bvh(victim)
    ctx.make_trace_lights(victim)
    for name, _, _ in yt.FlatScene.POOLS:
        getattr(victim, name)[...] = 0x55  # whatever is still to be read would read this
    for name in ["cameras", "instances", "environments", "shapes", "textures", "materials"]:
        getattr(victim, name).view("u1")[...] = 0xAA
    del victim
    gc.collect()
    got = P.gpu_render(ctx, flat, params)  # (`flat` only supplies the camera's image size here)
    ctx.close()
    P.assert_identical(want, got, "render after the source arrays died")


@pytest.mark.parametrize("res", [64, 1000, 2600])
def test_state_round_trip_is_exact_for_every_chunking(res):
    """state_upload / state_download / get_image of frames below one bounce chunk, a few chunks and
    tens of chunks (2600 x 1733 px: 72 MB of image) with sizes that are not multiples of 4 MiB."""
    flat = P.SCENES["cornellbox"]()
    params = yt.trace_params(sampler="eyelight", resolution=res, samples=1, batch=1)
    ctx = P.gpu_context(flat)
    ctx.make_trace_state(flat, params)
    n = ctx.npixels
    rng = np.random.default_rng(res)
    image = rng.random((n, 4), dtype=np.float32)
    albedo = rng.random((n, 3), dtype=np.float32)
    normal = rng.random((n, 3), dtype=np.float32)
    hits = rng.integers(0, 1 << 30, n, dtype=np.int32)
    rngs = rng.integers(0, 1 << 62, (n, 2), dtype=np.uint64)
    ctx.upload_state(image, albedo, normal, hits, rngs, samples=7)
    st = ctx.download_state()
    assert st["samples"] == 7
    for k, a in [("image", image), ("albedo", albedo), ("normal", normal), ("hits", hits), ("rngs", rngs)]:
        assert st[k].tobytes() == a.tobytes(), k
    assert ctx.get_image().tobytes() == image.tobytes()
    ctx.close()


def test_intersect_batch_of_many_chunks():
    """6,000,000 rays up (192 MB) and hits down (144 MB) through the 16 MiB ring: same hit records as
    the same rays sent in small pieces."""
    flat = P.SCENES["cornellbox"]()
    ctx = P.gpu_context(flat)
    rays = np.tile(P.random_rays(flat, 60000), 100)
    whole = ctx.intersect_batch(rays)
    piece = ctx.intersect_batch(rays[:60000])
    assert whole[:60000].tobytes() == piece.tobytes()
    assert whole[-60000:].tobytes() == piece.tobytes()
    ctx.close()


def test_fresh_processes_that_open_with_the_large_uploads():
    """Six fresh processes in a row do what round 2's sacrificial preflight did (context, upload of
    the 1M-triangle plane, device BVH build, two batches, download): every one must finish."""
    body = os.path.join(ROOT, "tools", "first_process_body.py")
    for k in range(6):
        r = subprocess.run([sys.executable, body], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, f"process {k}: rc {r.returncode}\n{r.stderr[-2000:]}"
        assert "done" in r.stdout
