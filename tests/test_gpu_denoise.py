"""SURVEY.md §8(f) rank 3 — the denoiser slot (denoise_image, yocto_trace.cpp:1794-1872).
The reference's filter is OIDN (not vendored) or a copy, so there is no reference output;
the device filter (csrc/yt_denoise.h) is checked against its numpy restatement
(tests/denoise_check.py, 2e-5 relative: the two differ only in `exp`), by properties, and
by what a denoiser is for: a 4 spp render filtered is closer to the converged image."""
import numpy as np
import pytest

import parity as P
import denoise_check as D

yt = P.yt
pytestmark = pytest.mark.gpu


def _close(a, b, rel=2e-5):
    scale = np.maximum(np.abs(a), np.abs(b)) + 1e-6
    return float(np.max(np.abs(a - b) / scale)) <= rel


def _random_inputs(h, w, seed):
    r = np.random.default_rng(seed)
    albedo = r.uniform(0, 1, (h, w, 3)).astype("f4")
    albedo[: h // 3] = albedo[0, 0]          # a flat region
    albedo[h // 3:, : w // 4] = 0.0          # albedo below the demodulation floor
    normal = r.normal(size=(h, w, 3)).astype("f4")
    normal /= np.linalg.norm(normal, axis=-1, keepdims=True)
    normal[: h // 2] = (0, 0, 1)
    render = np.empty((h, w, 4), "f4")
    render[..., :3] = albedo * 0.8 + r.exponential(0.3, (h, w, 3)).astype("f4")
    render[..., 3] = r.uniform(0, 1, (h, w)).astype("f4")
    return render, albedo, normal


@pytest.mark.parametrize("h,w,kw", [(37, 53, {}), (64, 129, dict(levels=3, sigma_color=1.5)),
                                    (5, 3, {}), (1, 1, {}), (40, 40, dict(levels=0))])
def test_device_filter_equals_its_numpy_restatement(h, w, kw):
    render, albedo, normal = _random_inputs(h, w, 7 + h)
    ctx = yt.Context(0)
    try:
        got = ctx.denoise_image(render, albedo, normal, **kw)
    finally:
        ctx.close()
    want = D.denoise(render, albedo, normal, **kw)
    assert got.shape == want.shape
    assert np.array_equal(got[..., 3], render[..., 3])  # alpha passes through
    assert _close(got[..., :3], want[..., :3]), float(np.max(np.abs(got - want)))


@pytest.mark.parametrize("h,w,levels", [(37, 53, 5), (64, 129, 5), (200, 333, 6), (9, 70, 4), (1080 // 4, 1920 // 4, 5)])
def test_lds_tiled_levels_equal_the_untiled_kernel(h, w, levels, monkeypatch):
    """k_atrous_lds (residue-class tiles staged through LDS, spacings <= 16) runs the same taps
    in the same order as k_atrous: byte-identical output, ragged sizes, a level beyond 16."""
    render, albedo, normal = _random_inputs(h, w, 3 * h + w)
    out = []
    for simple in ["0", "1"]:
        monkeypatch.setenv("YTHIP_DENOISE_SIMPLE", simple)
        ctx = yt.Context(0)
        try:
            out.append(ctx.denoise_image(render, albedo, normal, levels=levels))
        finally:
            ctx.close()
    assert out[0].tobytes() == out[1].tobytes()


def test_properties():
    ctx = yt.Context(0)
    try:
        h, w = 48, 80
        flat_alb = np.full((h, w, 3), 0.5, "f4")
        flat_nrm = np.tile(np.array([0, 1, 0], "f4"), (h, w, 1))
        const = np.full((h, w, 4), 0.25, "f4")
        out = ctx.denoise_image(const, flat_alb, flat_nrm)
        assert np.allclose(out, const, rtol=1e-6, atol=0)           # a constant image is a fixed point
        # with flat guides and a loose colour tolerance the filter is a weighted mean:
        # min <= out <= max, and the noise goes down
        r = np.random.default_rng(3)
        noisy = const.copy()
        noisy[..., :3] += r.normal(0, 0.05, (h, w, 3)).astype("f4")
        out = ctx.denoise_image(noisy, flat_alb, flat_nrm, sigma_color=10.0)
        assert out[..., :3].min() >= noisy[..., :3].min() - 1e-6 and out[..., :3].max() <= noisy[..., :3].max() + 1e-6
        assert out[..., :3].std() < 0.2 * noisy[..., :3].std()
        # a normal discontinuity is not crossed: two half planes with different colours stay apart
        two = const.copy()
        two[:, w // 2:, :3] = 0.75
        nrm = flat_nrm.copy()
        nrm[:, w // 2:] = (1, 0, 0)
        out = ctx.denoise_image(two, flat_alb, nrm, sigma_color=100.0)
        assert np.allclose(out[:, : w // 2, :3], 0.25, atol=1e-4) and np.allclose(out[:, w // 2:, :3], 0.75, atol=1e-4)
        # texture detail lives in the albedo: demodulation keeps it exactly under flat lighting
        alb = r.uniform(0.2, 1, (h, w, 3)).astype("f4")
        lit = np.concatenate([alb * np.float32(0.5), np.ones((h, w, 1), "f4")], -1)
        out = ctx.denoise_image(lit, alb, flat_nrm, sigma_albedo=100.0, sigma_color=100.0)
        assert np.allclose(out, lit, rtol=1e-5, atol=1e-7)
        # argument checks
        with pytest.raises(yt.YthipError):
            ctx.denoise_image(const, flat_alb, flat_nrm, levels=99)
        with pytest.raises(yt.YthipError):
            ctx.denoise_image(const, flat_alb, flat_nrm, sigma_color=0.0)
    finally:
        ctx.close()


def test_resident_state_hand_off_and_what_it_is_for():
    """The hand-off of yocto_trace.cpp:1615-1618 on the resident state: same result as the
    host-buffer entry on the downloaded state, invalidated by the next batch; and a 4 spp
    Cornell box filtered is closer (RMSE on the tonemapped image) to the 1024 spp render."""
    flat = P.SCENES["cornellbox"]()
    ctx = P.gpu_context(flat)
    try:
        p = yt.trace_params(sampler="path", resolution=256, samples=4, batch=4)
        st = P.gpu_render(ctx, flat, p)
        h, w = st["height"], st["width"]
        den = ctx.denoise_state()
        ptr = yt.C.c_void_p()
        assert ctx.lib.ythip_state_device_denoised(ctx.h, yt.C.byref(ptr)) == 0 and ptr.value
        again = ctx.denoise_image(st["image"].reshape(h, w, 4), st["albedo"].reshape(h, w, 3),
                                  st["normal"].reshape(h, w, 3))
        assert den.tobytes() == again.tobytes()
        den = ctx.denoise_state()  # (the host-buffer call reused the working images)
        pr = yt.trace_params(sampler="path", resolution=256, samples=1024, batch=1024)
        ref = P.gpu_render(ctx, flat, pr)["image"].reshape(h, w, 4)
        assert ctx.lib.ythip_state_device_denoised(ctx.h, yt.C.byref(ptr)) != 0  # the state moved on

        def ldr(x):
            return np.clip(x[..., :3], 0, 1) ** (1 / 2.2)
        noisy = st["image"].reshape(h, w, 4)
        e_noisy = float(np.sqrt(np.mean((ldr(noisy) - ldr(ref)) ** 2)))
        e_den = float(np.sqrt(np.mean((ldr(den) - ldr(ref)) ** 2)))
        print(f"RMSE vs 1024 spp: 4 spp {e_noisy:.4f}, 4 spp denoised {e_den:.4f}")
        assert e_den < 0.5 * e_noisy
        # a sliced state is refused
        ctx.make_trace_state(flat, p, cols=(0, 2))
        with pytest.raises(yt.YthipError):
            ctx.denoise_state()
        ctx.make_trace_state(flat, p, rows=(0, h // 2))
        with pytest.raises(yt.YthipError):
            ctx.denoise_state()
    finally:
        ctx.close()
