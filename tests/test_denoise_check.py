"""CPU checks of tests/denoise_check.py (the numpy restatement the GPU tests compare the
device denoiser with): a filter that is wrong in the restatement would make the GPU
equality test meaningless."""
import numpy as np

import denoise_check as D


def test_constant_image_is_a_fixed_point_and_alpha_passes_through():
    h, w = 12, 17
    img = np.full((h, w, 4), 0.3, "f4")
    img[..., 3] = np.linspace(0, 1, h * w, dtype="f4").reshape(h, w)
    out = D.denoise(img, np.full((h, w, 3), 0.6, "f4"), np.tile(np.array([0, 0, 1], "f4"), (h, w, 1)))
    assert np.allclose(out[..., :3], 0.3, rtol=1e-6) and np.array_equal(out[..., 3], img[..., 3])


def test_levels_zero_is_demodulate_remodulate():
    r = np.random.default_rng(0)
    img = r.uniform(0, 2, (9, 9, 4)).astype("f4")
    alb = r.uniform(0, 1, (9, 9, 3)).astype("f4")
    nrm = r.normal(size=(9, 9, 3)).astype("f4")
    out = D.denoise(img, alb, nrm, levels=0)
    den = np.maximum(alb, np.float32(0.01))
    assert np.array_equal(out[..., :3], (img[..., :3] / den) * den)


def test_guides_stop_the_blur_and_noise_goes_down():
    r = np.random.default_rng(1)
    h, w = 32, 48
    img = np.full((h, w, 4), 0.25, "f4")
    img[:, w // 2:, :3] = 0.75
    nrm = np.tile(np.array([0, 1, 0], "f4"), (h, w, 1))
    nrm[:, w // 2:] = (1, 0, 0)
    alb = np.full((h, w, 3), 0.5, "f4")
    noisy = img.copy()
    noisy[..., :3] += r.normal(0, 0.03, (h, w, 3)).astype("f4")
    out = D.denoise(noisy, alb, nrm, sigma_color=100.0)
    assert abs(out[:, : w // 2, :3].mean() - 0.25) < 0.01 and abs(out[:, w // 2:, :3].mean() - 0.75) < 0.01
    assert out[:, : w // 2, :3].std() < 0.3 * noisy[:, : w // 2, :3].std()
