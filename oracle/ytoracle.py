"""ctypes access to oracle/libyt_oracle.so — the plain-CPU RESTATEMENT of the
reference's hot path (oracle/yt_oracle.cpp).  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import this."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "yocto-gl_amd"))
import ythip as yt  # noqa: E402  (POD struct definitions only; no library call)

LIB_PATH = os.path.join(HERE, "libyt_oracle.so")
_lib = None


class OracleError(RuntimeError):
    pass


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise OracleError(f"{LIB_PATH} not built (make -C oracle oracle)")
        L = C.CDLL(LIB_PATH)
        L.yto_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc:
        raise OracleError(lib().yto_last_error().decode())


def make_rngs(seed, n):
    out = np.zeros((n, 2), "u8")
    _check(lib().yto_make_rngs(C.c_uint64(seed), C.c_int64(n), out.ctypes.data_as(C.c_void_p)))
    return out


def rand1f(seed, seq, n):
    state = np.zeros(2, "u8")
    out = np.zeros(n, "f4")
    _check(lib().yto_rand1f(C.c_uint64(seed), C.c_uint64(seq), n, state.ctypes.data_as(C.c_void_p),
                            out.ctypes.data_as(C.c_void_p)))
    return state, out


def state_size(camera, resolution):
    cam = np.ascontiguousarray(camera, yt.camera_dt).reshape(1)
    w, h = C.c_int(), C.c_int()
    _check(lib().yto_state_size(cam.ctypes.data_as(C.c_void_p), resolution, C.byref(w), C.byref(h)))
    return w.value, h.value


class Bundle:
    """A flat scene + the reference-layout BVH and lights (built by the host
    builders of libythip, which tests/test_host.py pins node-for-node against the
    reference's make_scene_bvh / make_trace_lights, or handed in)."""

    def __init__(self, flat, bvh=None, lights=None, highquality=False):
        self.flat = flat
        self.bvh = bvh if bvh is not None else yt.host_make_bvh(flat, highquality)
        self.lights = lights if lights is not None else yt.host_make_lights(flat)
        self._cs, self._cb, self._cl = flat.c_struct(), self.bvh.c_struct(), self.lights.c_struct()

    def supported(self, params):
        return lib().yto_supported(C.byref(self._cs), C.byref(self._cl), C.byref(params)) == 0

    def camera_rays(self, params, rngs=None):
        w, h = state_size(self.flat.cameras[params.camera], params.resolution)
        if rngs is None:
            rngs = make_rngs(params.seed, w * h)
        rngs = np.ascontiguousarray(rngs, "u8")
        rays = np.zeros(w * h, yt.ray_dt)
        _check(lib().yto_camera_rays(C.byref(self._cs), C.byref(params), w, h,
                                     rngs.ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.c_void_p)))
        return rays

    def intersect_batch(self, rays, find_any=False):
        rays = np.ascontiguousarray(rays, yt.ray_dt)
        hits = np.zeros(len(rays), yt.hit_dt)
        _check(lib().yto_intersect_batch(C.byref(self._cs), C.byref(self._cb),
                                         rays.ctypes.data_as(C.c_void_p), C.c_int64(len(rays)),
                                         int(find_any), hits.ctypes.data_as(C.c_void_p)))
        return hits

    def intersect_instance_batch(self, instances, rays, find_any=False):
        rays = np.ascontiguousarray(rays, yt.ray_dt)
        instances = np.ascontiguousarray(instances, "i4")
        hits = np.zeros(len(rays), yt.hit_dt)
        _check(lib().yto_intersect_instance_batch(
            C.byref(self._cs), C.byref(self._cb), instances.ctypes.data_as(C.c_void_p),
            rays.ctypes.data_as(C.c_void_p), C.c_int64(len(rays)), int(find_any),
            hits.ctypes.data_as(C.c_void_p)))
        return hits

    def render(self, params, state=None):
        """make_trace_state + trace_samples until params.samples (or continue
        from `state`, a dict as returned here)."""
        w, h = state_size(self.flat.cameras[params.camera], params.resolution)
        n = w * h
        if state is None:
            st = dict(image=np.zeros((n, 4), "f4"), albedo=np.zeros((n, 3), "f4"),
                      normal=np.zeros((n, 3), "f4"), hits=np.zeros(n, "i4"),
                      rngs=make_rngs(params.seed, n), samples=0)
        else:
            st = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v)
                  for k, v in state.items()}
        samples = C.c_int(st["samples"])
        while samples.value < params.samples:
            _check(lib().yto_trace_samples(
                C.byref(self._cs), C.byref(self._cb), C.byref(self._cl), C.byref(params), w, h,
                *[st[k].ctypes.data_as(C.c_void_p) for k in ["image", "albedo", "normal", "hits", "rngs"]],
                C.byref(samples)))
        st.update(samples=samples.value, width=w, height=h)
        return st
