"""The restated libm (yt_libm.h) as the DEVICE computes it, against the host's glibc: the
double-precision kernels use v_fma_f64 where glibc's FMA build uses vfmadd*sd, the fdlibm
float functions plain IEEE float arithmetic — the results must be the same bits."""
import ctypes as C

import numpy as np
import pytest

import parity as P
import ythip as yt

pytestmark = pytest.mark.gpu
libm = C.CDLL("libm.so.6")


def _host(fn, x, y=None):
    f = getattr(libm, fn)
    f.restype = C.c_float
    f.argtypes = [C.c_float] * (1 if y is None else 2)
    if y is None:
        return np.array([f(float(a)) for a in x], np.float32)
    return np.array([f(float(a), float(b)) for a, b in zip(x, y)], np.float32)


def _same(a, b):
    ai, bi = a.view(np.uint32), b.view(np.uint32)
    return (ai == bi) | (np.isnan(a) & np.isnan(b))


def _args(n, seed):
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)  # every exponent
    small = rng.uniform(-8, 8, n).astype(np.float32)                                         # the tracer's range
    unit = rng.uniform(-1, 1, n).astype(np.float32)
    return np.concatenate([bits, small, unit, np.array([0, -0.0, 1, -1, np.inf, -np.inf, np.nan, 1e-45, 88.7, -104,
                                                        120, 1e10, 3.4e38], np.float32)])


@pytest.mark.parametrize("fn", ["sinf", "cosf", "expf", "exp2f", "logf", "atanf", "acosf"])
def test_device_libm_one_argument(fn):
    flat = P.SCENES["cornellbox"]()
    ctx = yt.Context(0)
    x = _args(60_000, 5)
    got = ctx.test_libm(fn, x)
    ctx.close()
    ok = _same(got, _host(fn, x))
    assert ok.all(), (fn, x[~ok][:5], got[~ok][:5])


@pytest.mark.parametrize("fn", ["atan2f", "powf", "fmodf"])
def test_device_libm_two_arguments(fn):
    ctx = yt.Context(0)
    x, y = _args(60_000, 7), _args(60_000, 11)
    if fn == "powf":  # half of the pairs where the tracer uses it: bases in (0, 2), exponents in (0, 8)
        with np.errstate(all="ignore"):
            x[::2] = np.abs(x[::2]) % 2
            y[::2] = np.abs(y[::2]) % 8
    got = ctx.test_libm(fn, x, y)
    ctx.close()
    ok = _same(got, _host(fn, x, y))
    assert ok.all(), (fn, x[~ok][:5], y[~ok][:5], got[~ok][:5])


def test_device_ieee_division_and_sqrt():
    ctx = yt.Context(0)
    x, y = _args(60_000, 13), _args(60_000, 17)
    with np.errstate(all="ignore"):
        assert _same(ctx.test_libm("div", x, y), x / y).all()
        assert _same(ctx.test_libm("sqrtf", x), np.sqrt(x)).all()
    ctx.close()


def test_host_libm_agreement_probe():
    """ythip_host_libm_matches: on this image (glibc 2.35, a CPU with FMA) the host's libm IS the one the
    device restates, so the probe must say yes; the shim prints a warning where it says no."""
    ctx = yt.Context(0)
    assert ctx.host_libm_matches()
    ctx.close()
