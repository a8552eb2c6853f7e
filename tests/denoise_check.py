"""numpy restatement of csrc/yt_denoise.h (TEST INFRASTRUCTURE) — the à-trous filter as its
header defines it, float32 throughout, taps and sums in the kernel's order.  The only
difference from the device is `exp` (numpy's vs v_exp_f32), hence the 2e-5 tolerance of the
tests that use it."""
import numpy as np

f32 = np.float32
KERN = np.array([3 / 8, 1 / 4, 1 / 16], f32)
DEFAULTS = dict(levels=5, sigma_color=4.0, sigma_normal=0.35, sigma_albedo=0.1)


def _dist2(a, b):
    d = a - b
    return d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]


def _lum(c):
    return (c[..., 0] + c[..., 1] + c[..., 2]) / f32(3)


def denoise(render, albedo, normal, **kw):
    p = dict(DEFAULTS, **kw)
    render, albedo, normal = (np.asarray(x, f32) for x in (render, albedo, normal))
    h, w = render.shape[:2]
    inv_sn2 = f32(1) / (f32(p["sigma_normal"]) * f32(p["sigma_normal"]))
    inv_sa2 = f32(1) / (f32(p["sigma_albedo"]) * f32(p["sigma_albedo"]))
    inv_sc2 = f32(1) / (f32(p["sigma_color"]) * f32(p["sigma_color"]))
    den = np.maximum(albedo, f32(0.01))
    irr = render[..., :3] / den
    scale = f32(1)
    ys, xs = np.mgrid[0:h, 0:w]
    for level in range(p["levels"]):
        step = 1 << level
        acc = np.zeros((h, w, 3), f32)
        wsum = np.zeros((h, w), f32)
        lp = _lum(irr)
        c_l = inv_sc2 * scale
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                qy, qx = ys + dy * step, xs + dx * step
                ok = (qy >= 0) & (qy < h) & (qx >= 0) & (qx < w)
                qy, qx = np.clip(qy, 0, h - 1), np.clip(qx, 0, w - 1)
                cq, nq, aq = irr[qy, qx], normal[qy, qx], albedo[qy, qx]
                m = np.maximum(lp, _lum(cq))
                dc = _dist2(irr, cq) / (m * m + f32(1e-4))
                d = _dist2(normal, nq) * inv_sn2 + _dist2(albedo, aq) * inv_sa2 + dc * c_l
                wgt = (KERN[abs(dx)] * KERN[abs(dy)]) * np.exp(-d, dtype=f32)
                wgt = np.where(ok, wgt, f32(0)).astype(f32)
                acc = acc + wgt[..., None] * cq  # (adding +0 for a skipped tap leaves the sum as it was)
                wsum = wsum + wgt
        irr = acc / wsum[..., None]
        scale = scale * f32(4)
    out = np.empty((h, w, 4), f32)
    out[..., :3] = irr * den
    out[..., 3] = render[..., 3]
    return out
