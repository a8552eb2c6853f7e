"""TEST INFRASTRUCTURE — ctypes binding of oracle/_ref/libyocto_ref.so, the
UNMODIFIED reference (Yocto/GL) compiled with g++ by oracle/Makefile plus the
C-ABI harness oracle/ref_harness.cpp.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under yocto-gl_amd/ does.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(_HERE), "yocto-gl_amd"))
import ythip as yt  # noqa: E402  (POD type definitions only)

REF_LIB = os.path.join(_HERE, "_ref", "libyocto_ref.so")
_lib = None


def available():
    return os.path.exists(REF_LIB)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(f"{REF_LIB} missing: run `make -C oracle ref` where "
                               "/root/reference exists")
        L = C.CDLL(REF_LIB)
        vp = C.c_void_p
        sig = {
            "ref_scene_new": (vp, []),
            "ref_scene_free": (None, [vp]),
            "ref_scene_from_flat": (vp, [C.POINTER(yt.CScene)]),
            "ref_scene_cornellbox": (vp, []),
            "ref_add_recty": (C.c_int, [vp, C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_int]),
            "ref_add_rect": (C.c_int, [vp, C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_int]),
            "ref_add_uvsphere": (C.c_int, [vp, C.c_int, C.c_int, C.c_float, C.c_int]),
            "ref_add_sphere": (C.c_int, [vp, C.c_int, C.c_float, C.c_int]),
            "ref_add_hair": (C.c_int, [vp, C.c_int, C.c_int, C.c_int] + [C.c_float] * 4),
            "ref_add_random_points": (C.c_int, [vp, C.c_int] + [C.c_float] * 4),
            "ref_add_material": (C.c_int, [vp, vp]),
            "ref_add_instance": (C.c_int, [vp, vp, C.c_int, C.c_int]),
            "ref_add_environment": (C.c_int, [vp, vp] + [C.c_float] * 3 + [C.c_int]),
            "ref_add_camera_lookat": (C.c_int, [vp] + [C.c_float] * 10 + [C.c_int]),
            "ref_add_texture": (C.c_int, [vp] + [C.c_int] * 6 + [vp]),
            "ref_add_sunsky_texture": (C.c_int, [vp, C.c_int, C.c_int, C.c_float]),
            "ref_scene_commit": (None, [vp]),
            "ref_scene_flat": (None, [vp, C.POINTER(yt.CScene)]),
            "ref_bvh_build": (vp, [vp, C.c_int]),
            "ref_bvh_free": (None, [vp]),
            "ref_scene_set_vertices": (C.c_int, [vp, C.c_int, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64]),
            "ref_scene_set_instance_frame": (C.c_int, [vp, C.c_int, vp]),
            "ref_bvh_update": (C.c_double, [vp, vp, vp, C.c_int, vp, C.c_int]),
            "ref_bvh_flat": (None, [vp, C.POINTER(yt.CBvh)]),
            "ref_lights_build": (vp, [vp]),
            "ref_lights_free": (None, [vp]),
            "ref_lights_flat": (None, [vp, C.POINTER(yt.CLights)]),
            "ref_state_make": (vp, [vp, C.POINTER(yt.CParams)]),
            "ref_state_free": (None, [vp]),
            "ref_state_info": (None, [vp] + [C.POINTER(C.c_int)] * 3),
            "ref_state_get": (None, [vp] + [vp] * 5),
            "ref_state_set": (None, [vp] + [vp] * 5 + [C.c_int]),
            "ref_trace_samples": (C.c_double, [vp, vp, vp, vp, C.POINTER(yt.CParams)]),
            "ref_intersect_batch": (None, [vp, vp, vp, C.c_int64, C.c_int, vp]),
            "ref_intersect_instance_batch": (None, [vp, vp, vp, vp, C.c_int64, C.c_int, vp]),
            "ref_camera_rays": (None, [vp, vp, C.POINTER(yt.CParams), vp]),
            "ref_trace_sample": (None, [vp, vp, vp, vp, C.POINTER(yt.CParams), C.c_int, C.c_int, C.c_int]),
            "ref_guide_image": (None, [vp, C.c_int, vp]),
            "ref_eval_shading": (None, [vp, vp, vp, C.c_int64, vp]),
            "ref_eval_environment": (None, [vp, vp, C.c_int64, vp]),
            "ref_scene_load": (vp, [C.c_char_p]),
            "ref_scene_save": (C.c_int, [vp, C.c_char_p]),
            "ref_image_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), vp]),
            "ref_load_error": (C.c_char_p, []),
            "ref_tonemap": (None, [vp, C.c_int64, C.c_float, C.c_int, C.c_int, vp, vp]),
            "ref_make_rng": (None, [C.c_uint64, C.c_uint64, vp]),
            "ref_rand1f": (None, [vp, C.c_int, vp]),
            "ref_hardware_concurrency": (C.c_int, []),
            "ref_params_save": (C.c_int, [C.POINTER(yt.CParams), C.c_char_p]),
            "ref_params_load": (C.c_int, [C.c_char_p, C.POINTER(yt.CParams)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class RefScene:
    """scene_data living inside the reference library."""

    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def cornellbox():
        return RefScene(lib().ref_scene_cornellbox())

    @staticmethod
    def new():
        return RefScene(lib().ref_scene_new())

    @staticmethod
    def load(filename):
        """The reference's own load_scene (+ tesselate_subdivs, as ytrace does)."""
        h = lib().ref_scene_load(str(filename).encode())
        if not h:
            raise RuntimeError(lib().ref_load_error().decode())
        return RefScene(h)

    def save(self, filename):
        """The reference's own save_scene: JSON + shapes/*.ply (+ textures)."""
        if lib().ref_scene_save(self.h, str(filename).encode()):
            raise RuntimeError(lib().ref_load_error().decode())

    @staticmethod
    def from_flat(flat):
        cs = flat.c_struct()
        return RefScene(lib().ref_scene_from_flat(C.byref(cs)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_scene_free(self.h)
            self.h = None

    def add_recty(self, steps, scale, uvscale=(1, 1), triangulate=True):
        return lib().ref_add_recty(self.h, steps[0], steps[1], scale[0], scale[1],
                                   uvscale[0], uvscale[1], int(triangulate))

    def add_rect(self, steps, scale, uvscale=(1, 1), triangulate=True):
        return lib().ref_add_rect(self.h, steps[0], steps[1], scale[0], scale[1],
                                  uvscale[0], uvscale[1], int(triangulate))

    def add_uvsphere(self, steps, scale, triangulate=True):
        return lib().ref_add_uvsphere(self.h, steps[0], steps[1], scale, int(triangulate))

    def add_sphere(self, steps, scale, triangulate=False):
        return lib().ref_add_sphere(self.h, steps, scale, int(triangulate))

    def add_hair(self, base_shape, steps, length, radius):
        return lib().ref_add_hair(self.h, base_shape, steps[0], steps[1], length[0],
                                  length[1], radius[0], radius[1])

    def add_random_points(self, num, size, radius):
        return lib().ref_add_random_points(self.h, num, size[0], size[1], size[2], radius)

    def add_material(self, **kw):
        tmp = yt.FlatScene()
        tmp.add_material(**kw)
        m = np.ascontiguousarray(tmp.materials)
        return lib().ref_add_material(self.h, m.ctypes.data)

    def add_instance(self, shape, material, frame=yt.IDENTITY_FRAME):
        f = np.ascontiguousarray(frame, "f4").reshape(12)
        return lib().ref_add_instance(self.h, f.ctypes.data, shape, material)

    def add_environment(self, emission=(1, 1, 1), emission_tex=-1,
                        frame=yt.IDENTITY_FRAME):
        f = np.ascontiguousarray(frame, "f4").reshape(12)
        return lib().ref_add_environment(self.h, f.ctypes.data, emission[0],
                                         emission[1], emission[2], emission_tex)

    def add_camera_lookat(self, eye, center, lens=0.05, film=0.036, aspect=1.5,
                          aperture=0.0, orthographic=False):
        return lib().ref_add_camera_lookat(self.h, eye[0], eye[1], eye[2], center[0],
                                           center[1], center[2], lens, film, aspect,
                                           aperture, int(orthographic))

    def add_texture(self, pixels, linear=None, nearest=False, clamp=False):
        pixels = np.ascontiguousarray(pixels)
        is_float = pixels.dtype != np.uint8
        if is_float:
            pixels = np.ascontiguousarray(pixels, "f4")
        h, w = pixels.shape[:2]
        linear = is_float if linear is None else linear
        return lib().ref_add_texture(self.h, w, h, int(linear), int(nearest),
                                     int(clamp), int(is_float), pixels.ctypes.data)

    def add_sunsky_texture(self, width, height, sun_angle):
        return lib().ref_add_sunsky_texture(self.h, width, height, sun_angle)

    def set_vertices(self, shape, positions=None, normals=None, radius=None):
        """Edit a shape's vertices in place (element lists untouched)."""
        p = None if positions is None else np.ascontiguousarray(positions, "f4").reshape(-1, 3)
        n = None if normals is None else np.ascontiguousarray(normals, "f4").reshape(-1, 3)
        r = None if radius is None else np.ascontiguousarray(radius, "f4").reshape(-1)
        ptr = lambda a: None if a is None or a.size == 0 else a.ctypes.data  # noqa: E731
        cnt = lambda a: 0 if a is None else len(a)  # noqa: E731
        rc = lib().ref_scene_set_vertices(self.h, int(shape), ptr(p), cnt(p), ptr(n), cnt(n), ptr(r), cnt(r))
        if rc != 0:
            raise ValueError("set_vertices: shape / count mismatch")

    def set_instance_frame(self, instance, frame):
        f = np.ascontiguousarray(frame, "f4").reshape(12)
        if lib().ref_scene_set_instance_frame(self.h, int(instance), f.ctypes.data) != 0:
            raise ValueError("set_instance_frame: instance out of range")

    def flat(self):
        """Deep copy of the scene as a FlatScene (numpy-owned)."""
        lib().ref_scene_commit(self.h)
        cs = yt.CScene()
        lib().ref_scene_flat(self.h, C.byref(cs))
        return yt.FlatScene.from_c(cs)


class RefBvh:
    def __init__(self, scene, highquality=False):
        self.h = lib().ref_bvh_build(scene.h, int(highquality))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_bvh_free(self.h)
            self.h = None

    def update(self, scene, updated_instances=(), updated_shapes=()):
        """update_scene_bvh (yocto_bvh.cpp:434-451); returns the seconds it took."""
        vi = np.ascontiguousarray(list(updated_instances), "i4")
        vs = np.ascontiguousarray(list(updated_shapes), "i4")
        return lib().ref_bvh_update(self.h, scene.h, vi.ctypes.data, len(vi), vs.ctypes.data, len(vs))

    def flat(self):
        cb = yt.CBvh()
        lib().ref_bvh_flat(self.h, C.byref(cb))
        return yt.FlatBvh.from_c(cb)


class RefLights:
    def __init__(self, scene):
        self.h = lib().ref_lights_build(scene.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_lights_free(self.h)
            self.h = None

    def flat(self):
        cl = yt.CLights()
        lib().ref_lights_flat(self.h, C.byref(cl))
        return yt.FlatLights.from_c(cl)


class RefState:
    def __init__(self, scene, params):
        self.h = lib().ref_state_make(scene.h, C.byref(params))
        w, h, s = C.c_int(), C.c_int(), C.c_int()
        lib().ref_state_info(self.h, C.byref(w), C.byref(h), C.byref(s))
        self.width, self.height = w.value, h.value

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_state_free(self.h)
            self.h = None

    @property
    def samples(self):
        w, h, s = C.c_int(), C.c_int(), C.c_int()
        lib().ref_state_info(self.h, C.byref(w), C.byref(h), C.byref(s))
        return s.value

    def get(self):
        n = self.width * self.height
        image = np.zeros((n, 4), "f4")
        albedo = np.zeros((n, 3), "f4")
        normal = np.zeros((n, 3), "f4")
        hits = np.zeros(n, "i4")
        rngs = np.zeros((n, 2), "u8")
        lib().ref_state_get(self.h, image.ctypes.data, albedo.ctypes.data,
                            normal.ctypes.data, hits.ctypes.data, rngs.ctypes.data)
        return dict(image=image, albedo=albedo, normal=normal, hits=hits, rngs=rngs,
                    samples=self.samples, width=self.width, height=self.height)

    def set(self, image=None, albedo=None, normal=None, hits=None, rngs=None,
            samples=0):
        arrs = [None if a is None else np.ascontiguousarray(a, dt)
                for a, dt in [(image, "f4"), (albedo, "f4"), (normal, "f4"),
                              (hits, "i4"), (rngs, "u8")]]
        lib().ref_state_set(self.h, *[None if a is None else a.ctypes.data
                                      for a in arrs], samples)


def trace_samples(state, scene, bvh, lights, params):
    """The reference's trace_samples; returns wall seconds."""
    return lib().ref_trace_samples(state.h, scene.h, bvh.h, lights.h, C.byref(params))


def load_image(filename):
    """The reference's load_image: [h, w, 4] float32 (what its apps wrote)."""
    w, h = C.c_int(), C.c_int()
    fn = str(filename).encode()
    if lib().ref_image_load(fn, C.byref(w), C.byref(h), None):
        raise RuntimeError(lib().ref_load_error().decode())
    out = np.zeros((h.value, w.value, 4), "f4")
    if lib().ref_image_load(fn, C.byref(w), C.byref(h), out.ctypes.data):
        raise RuntimeError(lib().ref_load_error().decode())
    return out


def trace_sample(state, scene, bvh, lights, params, i, j, sample):
    """The reference's trace_sample: one sample of pixel (i, j)."""
    lib().ref_trace_sample(state.h, scene.h, bvh.h, lights.h, C.byref(params), i, j, sample)


def guide_image(state, which):
    """The reference's get_albedo_image (which=0) / get_normal_image (which=1)."""
    out = np.zeros((state.height, state.width, 4), "f4")
    lib().ref_guide_image(state.h, which, out.ctypes.data)
    return out


def intersect_batch(bvh, scene, rays, find_any=False):
    rays = np.ascontiguousarray(rays, yt.ray_dt)
    hits = np.zeros(len(rays), yt.hit_dt)
    lib().ref_intersect_batch(bvh.h, scene.h, rays.ctypes.data, len(rays),
                              int(find_any), hits.ctypes.data)
    return hits


def intersect_instance_batch(bvh, scene, instances, rays, find_any=False):
    rays = np.ascontiguousarray(rays, yt.ray_dt)
    instances = np.ascontiguousarray(instances, "i4")
    hits = np.zeros(len(rays), yt.hit_dt)
    lib().ref_intersect_instance_batch(bvh.h, scene.h, instances.ctypes.data,
                                       rays.ctypes.data, len(rays), int(find_any),
                                       hits.ctypes.data)
    return hits


def camera_rays(state, scene, params):
    rays = np.zeros(state.width * state.height, yt.ray_dt)
    lib().ref_camera_rays(state.h, scene.h, C.byref(params), rays.ctypes.data)
    return rays


def eval_shading(scene, hits, outgoing):
    hits = np.ascontiguousarray(hits, yt.hit_dt)
    outgoing = np.ascontiguousarray(outgoing, "f4")
    out = np.zeros((len(hits), 25), "f4")
    lib().ref_eval_shading(scene.h, hits.ctypes.data, outgoing.ctypes.data,
                           len(hits), out.ctypes.data)
    return out


def eval_environment(scene, dirs):
    dirs = np.ascontiguousarray(dirs, "f4")
    out = np.zeros((len(dirs), 3), "f4")
    lib().ref_eval_environment(scene.h, dirs.ctypes.data, len(dirs), out.ctypes.data)
    return out


def tonemap(hdr, exposure=0.0, filmic=False, srgb=True):
    """The reference's tonemap_image on [n, 4] floats: (float [n, 4], bytes [n, 4])."""
    hdr = np.ascontiguousarray(hdr, "f4").reshape(-1, 4)
    ldr, ldrb = np.zeros_like(hdr), np.zeros(hdr.shape, "u1")
    lib().ref_tonemap(hdr.ctypes.data, len(hdr), exposure, int(filmic), int(srgb), ldr.ctypes.data,
                      ldrb.ctypes.data)
    return ldr, ldrb


def make_rng(seed, seq):
    out = np.zeros(2, "u8")
    lib().ref_make_rng(seed, seq, out.ctypes.data)
    return out


def rand1f(state, n):
    state = np.ascontiguousarray(state, "u8").copy()
    out = np.zeros(n, "f4")
    lib().ref_rand1f(state.ctypes.data, n, out.ctypes.data)
    return out, state


def hardware_concurrency():
    return lib().ref_hardware_concurrency()


def fnv1a64(data: bytes) -> int:
    """FNV-1a-64 (SURVEY.md §8c KATs)."""
    arr = np.frombuffer(data, "u1")
    h = 1469598103934665603
    for b in arr.tolist():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h
