"""Zero-bounce batches and the pixel pool (`-m gpu`): k_trace finishes a batch with bounces <= 0 in its prologue, tile
by tile, and never reaches the pool's queue — so such a batch must be launched plain whatever the pool mode says
(ythip.hip, enqueue_samples).  Forced pool with a handful of workgroups, bounces 0: every pixel gets its samples and
the state equals the reference's."""
import numpy as np
import pytest

import parity as P
from parity import yt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sampler", ["path", "naive"])
def test_zero_bounce_batches_are_launched_plain(sampler):
    flat = P.SCENES["cornellbox"]()
    params = yt.trace_params(sampler=sampler, resolution=120, samples=4, batch=2, bounces=0)
    ctx = P.gpu_context(flat)
    ctx.set_pixel_pool(2, 3)
    got = P.gpu_render(ctx, flat, params)
    ctx.close()
    assert got["samples"] == 4
    ref = P.gpu_context(flat)
    ref.set_pixel_pool(0)
    want = P.gpu_render(ref, flat, params)
    ref.close()
    P.assert_identical(want, got, f"{sampler} bounces=0")
    if P.have_ref():
        P.assert_identical(P.RefBundle(flat).render(params), got, f"{sampler} bounces=0 vs the reference")
