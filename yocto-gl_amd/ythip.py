"""ctypes binding of libythip.so (include/ythip.h) + numpy mirrors of its POD types.

This is plumbing for tests / bench.py / multi-GPU launch: the product is the
HIP library.  There is NO CPU fallback here: if the library is missing or the
GPU call fails, an exception is raised (YthipError).

The Python-level names mirror the reference's lower-level rendering API
(libs/yocto/yocto_trace.h:160-190): make_trace_bvh / make_trace_lights /
make_trace_state / trace_samples / get_image.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YTHIP_LIB") or os.path.join(_HERE, "csrc", "libythip.so")  # YTHIP_LIB: A/B builds of the same HIP library

# ----------------------------------------------------------------------------
# numpy dtypes == C structs of include/ythip.h
# ----------------------------------------------------------------------------
camera_dt = np.dtype([("frame", "f4", 12), ("orthographic", "i4"), ("lens", "f4"),
                      ("film", "f4"), ("aspect", "f4"), ("focus", "f4"),
                      ("aperture", "f4")])
instance_dt = np.dtype([("frame", "f4", 12), ("shape", "i4"), ("material", "i4")])
material_dt = np.dtype([("type", "i4"), ("emission", "f4", 3), ("color", "f4", 3),
                        ("roughness", "f4"), ("metallic", "f4"), ("ior", "f4"),
                        ("scattering", "f4", 3), ("scanisotropy", "f4"),
                        ("trdepth", "f4"), ("opacity", "f4"),
                        ("emission_tex", "i4"), ("color_tex", "i4"),
                        ("roughness_tex", "i4"), ("scattering_tex", "i4"),
                        ("normal_tex", "i4")])
environment_dt = np.dtype([("frame", "f4", 12), ("emission", "f4", 3),
                           ("emission_tex", "i4")])
texture_dt = np.dtype([("width", "i4"), ("height", "i4"), ("linear", "i4"),
                       ("nearest", "i4"), ("clamp", "i4"), ("is_float", "i4"),
                       ("offset", "i8")])
shape_dt = np.dtype([("points_offset", "i8"), ("lines_offset", "i8"),
                     ("triangles_offset", "i8"), ("quads_offset", "i8"),
                     ("positions_offset", "i8"), ("normals_offset", "i8"),
                     ("texcoords_offset", "i8"), ("colors_offset", "i8"),
                     ("radius_offset", "i8"),
                     ("num_points", "i4"), ("num_lines", "i4"),
                     ("num_triangles", "i4"), ("num_quads", "i4"),
                     ("num_positions", "i4"), ("num_normals", "i4"),
                     ("num_texcoords", "i4"), ("num_colors", "i4"),
                     ("num_radius", "i4"), ("pad_", "i4")])
node_dt = np.dtype([("bbox_min", "f4", 3), ("bbox_max", "f4", 3), ("start", "i4"),
                    ("num", "i2"), ("axis", "i1"), ("internal", "u1")])
light_dt = np.dtype([("instance", "i4"), ("environment", "i4"),
                     ("cdf_offset", "i8"), ("cdf_count", "i4"), ("pad_", "i4")])
hit_dt = np.dtype([("instance", "i4"), ("element", "i4"), ("u", "f4"), ("v", "f4"),
                   ("distance", "f4"), ("hit", "i4")])
ray_dt = np.dtype([("o", "f4", 3), ("d", "f4", 3), ("tmin", "f4"), ("tmax", "f4")])

assert camera_dt.itemsize == 72 and instance_dt.itemsize == 56
assert material_dt.itemsize == 84 and environment_dt.itemsize == 64
assert texture_dt.itemsize == 32 and shape_dt.itemsize == 112
assert node_dt.itemsize == 32 and light_dt.itemsize == 24
assert hit_dt.itemsize == 24 and ray_dt.itemsize == 32

IDENTITY_FRAME = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype="f4")
INVALIDID = -1

MATERIAL_TYPES = ["matte", "glossy", "reflective", "transparent", "refractive",
                  "subsurface", "volumetric", "gltfpbr"]
SAMPLERS = ["path", "pathdirect", "pathmis", "pathtest", "naive", "eyelight",
            "diagram", "furnace", "falsecolor"]
FALSECOLORS = ["position", "normal", "frontfacing", "gnormal", "gfrontfacing",
               "texcoord", "mtype", "color", "emission", "roughness", "opacity",
               "metallic", "delta", "instance", "shape", "material", "element",
               "highlight"]


# ----------------------------------------------------------------------------
# ctypes structs
# ----------------------------------------------------------------------------
class CDenoiseParams(C.Structure):
    _fields_ = [("levels", C.c_int32), ("sigma_color", C.c_float), ("sigma_normal", C.c_float),
                ("sigma_albedo", C.c_float)]


class CScene(C.Structure):
    _fields_ = [
        ("num_cameras", C.c_int32), ("num_instances", C.c_int32),
        ("num_environments", C.c_int32), ("num_shapes", C.c_int32),
        ("num_textures", C.c_int32), ("num_materials", C.c_int32),
        ("cameras", C.c_void_p), ("instances", C.c_void_p),
        ("environments", C.c_void_p), ("shapes", C.c_void_p),
        ("textures", C.c_void_p), ("materials", C.c_void_p),
        ("num_points", C.c_int64), ("num_lines", C.c_int64),
        ("num_triangles", C.c_int64), ("num_quads", C.c_int64),
        ("points", C.c_void_p), ("lines", C.c_void_p),
        ("triangles", C.c_void_p), ("quads", C.c_void_p),
        ("num_positions", C.c_int64), ("num_normals", C.c_int64),
        ("num_texcoords", C.c_int64), ("num_colors", C.c_int64),
        ("num_radius", C.c_int64),
        ("positions", C.c_void_p), ("normals", C.c_void_p),
        ("texcoords", C.c_void_p), ("colors", C.c_void_p),
        ("radius", C.c_void_p),
        ("num_pixelsf", C.c_int64), ("num_pixelsb", C.c_int64),
        ("pixelsf", C.c_void_p), ("pixelsb", C.c_void_p),
    ]


class CBvh(C.Structure):
    _fields_ = [("num_trees", C.c_int32), ("node_offset", C.c_void_p),
                ("prim_offset", C.c_void_p), ("nodes", C.c_void_p),
                ("primitives", C.c_void_p)]


class CLights(C.Structure):
    _fields_ = [("num_lights", C.c_int32), ("lights", C.c_void_p),
                ("num_cdf", C.c_int64), ("cdf", C.c_void_p)]


class CParams(C.Structure):
    """trace_params (yocto_trace.h:95-113), same defaults."""
    _fields_ = [("camera", C.c_int32), ("resolution", C.c_int32),
                ("sampler", C.c_int32), ("falsecolor", C.c_int32),
                ("samples", C.c_int32), ("bounces", C.c_int32),
                ("clamp", C.c_float), ("nocaustics", C.c_int32),
                ("envhidden", C.c_int32), ("tentfilter", C.c_int32),
                ("seed", C.c_uint64), ("embreebvh", C.c_int32),
                ("highqualitybvh", C.c_int32), ("noparallel", C.c_int32),
                ("pratio", C.c_int32), ("denoise", C.c_int32),
                ("batch", C.c_int32), ("fastmath", C.c_int32)]


TRACE_DEFAULT_SEED = 961748941


def trace_params(**kw):
    p = CParams(camera=0, resolution=1280, sampler=0, falsecolor=7, samples=512,
                bounces=8, clamp=10.0, nocaustics=0, envhidden=0, tentfilter=0,
                seed=TRACE_DEFAULT_SEED, embreebvh=0, highqualitybvh=0,
                noparallel=0, pratio=8, denoise=0, batch=1, fastmath=0)
    for k, v in kw.items():
        if k == "sampler" and isinstance(v, str):
            v = SAMPLERS.index(v)
        if k == "falsecolor" and isinstance(v, str):
            v = FALSECOLORS.index(v)
        if not hasattr(p, k):
            raise TypeError(f"unknown trace_params field {k!r}")
        setattr(p, k, v)
    return p


def params_to_json(params):
    """to_json(trace_params) — yocto_sceneio.cpp:5815-5833 (ythip_params_to_json)."""
    lib = load_library()
    n = lib.ythip_params_to_json(C.byref(params), None, 0)
    if n < 0:
        raise YthipError(lib.ythip_io_last_error().decode())
    buf = C.create_string_buffer(n + 1)
    lib.ythip_params_to_json(C.byref(params), buf, n + 1)
    return buf.value.decode()


def params_from_json(text, params=None):
    """from_json(trace_params): keys that are absent keep the values of `params` (default: trace_params{})."""
    lib = load_library()
    if params is None:
        params = CParams()
        lib.ythip_params_default(C.byref(params))
    b = text.encode() if isinstance(text, str) else text
    if lib.ythip_params_from_json(b, len(b), C.byref(params)):
        raise YthipError(lib.ythip_io_last_error().decode())
    return params


def load_ply(path, flip_texcoord=True):
    """load_shape (yocto_sceneio.cpp:1017-1033) of a PLY file through ythip_ply_open / ythip_ply_read:
    a dict of the shape's arrays (absent ones empty), converted straight from the mapped file."""
    lib = load_library()
    h = C.c_void_p()
    counts = np.zeros(1, shape_dt)
    if lib.ythip_ply_open(str(path).encode(), C.byref(h), counts.ctypes.data):
        raise YthipError(lib.ythip_io_last_error().decode())
    c = counts[0]
    out = {name: np.zeros((int(c["num_" + name]), n), dt) for name, dt, n in FlatScene.POOLS[:9]}
    ptr = lambda a: a.ctypes.data if a.size else None  # noqa: E731
    rc = lib.ythip_ply_read(h, int(flip_texcoord), ptr(out["positions"]), ptr(out["normals"]), ptr(out["texcoords"]),
                            ptr(out["colors"]), ptr(out["radius"]), ptr(out["points"]), ptr(out["lines"]),
                            ptr(out["triangles"]), ptr(out["quads"]))
    lib.ythip_ply_close(h)
    if rc:
        raise YthipError(lib.ythip_io_last_error().decode())
    return out


def load_scene_file(path, threads=0):
    """load_scene (yocto_sceneio.cpp:3618-3857) of a builtin-format scene through ythip_scene_open /
    ythip_scene_read into numpy pools: a FlatScene equal to the reference loader's scene_data flattened,
    plus the names ({"cameras": [...], ...}) and the index find_camera("") picks."""
    lib = load_library()
    h, counts = C.c_void_p(), CScene()
    if lib.ythip_scene_open(str(path).encode(), C.byref(h), C.byref(counts)):
        raise YthipError(lib.ythip_io_last_error().decode())
    fs = FlatScene()
    fs.cameras, fs.instances = np.zeros(counts.num_cameras, camera_dt), np.zeros(counts.num_instances, instance_dt)
    fs.environments, fs.shapes = np.zeros(counts.num_environments, environment_dt), np.zeros(counts.num_shapes, shape_dt)
    fs.textures, fs.materials = np.zeros(counts.num_textures, texture_dt), np.zeros(counts.num_materials, material_dt)
    for name, dt, n in FlatScene.POOLS:
        setattr(fs, name, np.zeros((getattr(counts, "num_" + name), n), dt))
    pools = fs.c_struct()
    rc = lib.ythip_scene_read(h, C.byref(pools), int(threads))
    names = {}
    for what, kind in enumerate(["cameras", "instances", "environments", "shapes", "textures", "materials"]):
        names[kind] = [lib.ythip_scene_name(h, what, k).decode() for k in range(len(getattr(fs, kind)))]
    camera = lib.ythip_scene_find_camera(h, b"")
    lib.ythip_scene_close(h)
    if rc:
        raise YthipError(lib.ythip_io_last_error().decode())
    return fs, names, camera


class CPoolInfo(C.Structure):
    _fields_ = [("mode", C.c_int32), ("workgroups", C.c_int32), ("decided", C.c_int32), ("on", C.c_int32),
                ("plain_ms_per_sample", C.c_float), ("pool_ms_per_sample", C.c_float)]


class CStreamInfo(C.Structure):
    _fields_ = [("ran", C.c_int32), ("generations", C.c_int32), ("launched", C.c_int32), ("bins", C.c_int32),
                ("groups", C.c_int32), ("path_slots", C.c_int32), ("rays", C.c_int64), ("lane_steps", C.c_int64), ("wave_steps", C.c_int64),
                ("finish_rays", C.c_int64), ("choice_state", C.c_int32), ("choice_streamed", C.c_int32),
                ("fused_ms_per_sample", C.c_float), ("stream_ms_per_sample", C.c_float)]


class CBuildInfo(C.Structure):
    _fields_ = [("device_trees", C.c_int32), ("host_trees", C.c_int32), ("fallbacks", C.c_int32),
                ("max_depth", C.c_int32), ("device_prims", C.c_int64), ("device_ms", C.c_double),
                ("build_ms", C.c_double), ("bake_ms", C.c_double), ("host_threads", C.c_int32),
                ("device_tlas", C.c_int32)]


class CStats(C.Structure):
    _fields_ = [("trace_launches", C.c_int64), ("trace_ms", C.c_double),
                ("rays", C.c_int64), ("nodes", C.c_int64),
                ("triangles", C.c_int64), ("quads", C.c_int64),
                ("lines", C.c_int64), ("points", C.c_int64),
                ("instances", C.c_int64), ("shades", C.c_int64),
                ("samples", C.c_int64)]


def _ptr(a):
    return None if a is None or a.size == 0 else a.ctypes.data


class FlatScene:
    """scene_data (yocto_scene.h:191-213) flattened into contiguous pools."""

    POOLS = [("points", "i4", 1), ("lines", "i4", 2), ("triangles", "i4", 3),
             ("quads", "i4", 4), ("positions", "f4", 3), ("normals", "f4", 3),
             ("texcoords", "f4", 2), ("colors", "f4", 4), ("radius", "f4", 1),
             ("pixelsf", "f4", 4), ("pixelsb", "u1", 4)]

    def __init__(self):
        self.cameras = np.zeros(0, camera_dt)
        self.instances = np.zeros(0, instance_dt)
        self.environments = np.zeros(0, environment_dt)
        self.shapes = np.zeros(0, shape_dt)
        self.textures = np.zeros(0, texture_dt)
        self.materials = np.zeros(0, material_dt)
        for name, dt, n in self.POOLS:
            setattr(self, name, np.zeros((0, n), dt))

    # -- authoring helpers ----------------------------------------------------
    def add_camera(self, frame, lens=0.05, film=0.036, aspect=1.5, focus=10000.0,
                   aperture=0.0, orthographic=False):
        c = np.zeros(1, camera_dt)
        c["frame"] = np.asarray(frame, "f4").reshape(12)
        c["orthographic"], c["lens"], c["film"] = int(orthographic), lens, film
        c["aspect"], c["focus"], c["aperture"] = aspect, focus, aperture
        self.cameras = np.concatenate([self.cameras, c])
        return len(self.cameras) - 1

    def add_material(self, type="matte", emission=(0, 0, 0), color=(0, 0, 0),
                     roughness=0.0, metallic=0.0, ior=1.5, scattering=(0, 0, 0),
                     scanisotropy=0.0, trdepth=0.01, opacity=1.0, emission_tex=-1,
                     color_tex=-1, roughness_tex=-1, scattering_tex=-1,
                     normal_tex=-1):
        m = np.zeros(1, material_dt)
        m["type"] = MATERIAL_TYPES.index(type) if isinstance(type, str) else type
        m["emission"], m["color"] = emission, color
        m["roughness"], m["metallic"], m["ior"] = roughness, metallic, ior
        m["scattering"], m["scanisotropy"] = scattering, scanisotropy
        m["trdepth"], m["opacity"] = trdepth, opacity
        m["emission_tex"], m["color_tex"] = emission_tex, color_tex
        m["roughness_tex"], m["scattering_tex"] = roughness_tex, scattering_tex
        m["normal_tex"] = normal_tex
        self.materials = np.concatenate([self.materials, m])
        return len(self.materials) - 1

    def add_instance(self, shape, material, frame=IDENTITY_FRAME):
        i = np.zeros(1, instance_dt)
        i["frame"] = np.asarray(frame, "f4").reshape(12)
        i["shape"], i["material"] = shape, material
        self.instances = np.concatenate([self.instances, i])
        return len(self.instances) - 1

    def add_instances(self, frames, shape, material):
        frames = np.asarray(frames, "f4").reshape(-1, 12)
        i = np.zeros(len(frames), instance_dt)
        i["frame"], i["shape"], i["material"] = frames, shape, material
        self.instances = np.concatenate([self.instances, i])

    def add_environment(self, emission=(1, 1, 1), emission_tex=-1,
                        frame=IDENTITY_FRAME):
        e = np.zeros(1, environment_dt)
        e["frame"] = np.asarray(frame, "f4").reshape(12)
        e["emission"], e["emission_tex"] = emission, emission_tex
        self.environments = np.concatenate([self.environments, e])
        return len(self.environments) - 1

    def add_texture(self, pixels, linear=None, nearest=False, clamp=False):
        pixels = np.asarray(pixels)
        h, w = pixels.shape[:2]
        is_float = pixels.dtype != np.uint8
        t = np.zeros(1, texture_dt)
        t["width"], t["height"] = w, h
        t["linear"] = int(is_float if linear is None else linear)
        t["nearest"], t["clamp"], t["is_float"] = int(nearest), int(clamp), int(is_float)
        if is_float:
            t["offset"] = len(self.pixelsf)
            self.pixelsf = np.concatenate(
                [self.pixelsf, pixels.astype("f4").reshape(-1, 4)])
        else:
            t["offset"] = len(self.pixelsb)
            self.pixelsb = np.concatenate([self.pixelsb, pixels.reshape(-1, 4)])
        self.textures = np.concatenate([self.textures, t])
        return len(self.textures) - 1

    def add_shape(self, positions, points=None, lines=None, triangles=None,
                  quads=None, normals=None, texcoords=None, colors=None,
                  radius=None):
        s = np.zeros(1, shape_dt)

        def put(name, arr, dt, n):
            if arr is None or len(arr) == 0:
                s[name + "_offset"], s["num_" + name] = -1, 0
                return
            arr = np.ascontiguousarray(arr, dt).reshape(-1, n)
            pool = getattr(self, name)
            s[name + "_offset"], s["num_" + name] = len(pool), len(arr)
            setattr(self, name, np.concatenate([pool, arr]))

        put("points", points, "i4", 1)
        put("lines", lines, "i4", 2)
        put("triangles", triangles, "i4", 3)
        put("quads", quads, "i4", 4)
        put("positions", positions, "f4", 3)
        put("normals", normals, "f4", 3)
        put("texcoords", texcoords, "f4", 2)
        put("colors", colors, "f4", 4)
        put("radius", radius, "f4", 1)
        self.shapes = np.concatenate([self.shapes, s])
        return len(self.shapes) - 1

    def shape_arrays(self, k):
        """Shape k's arrays as a dict of views (shape-local indices)."""
        s = self.shapes[k]
        out = {}
        for name in ["points", "lines", "triangles", "quads", "positions",
                     "normals", "texcoords", "colors", "radius"]:
            off, n = int(s[name + "_offset"]), int(s["num_" + name])
            out[name] = getattr(self, name)[off:off + n] if off >= 0 else None
        return out

    # -- C view -----------------------------------------------------------------
    def c_struct(self):
        for name in ["cameras", "instances", "environments", "shapes", "textures",
                     "materials"] + [p[0] for p in self.POOLS]:
            setattr(self, name, np.ascontiguousarray(getattr(self, name)))
        cs = CScene()
        cs.num_cameras, cs.num_instances = len(self.cameras), len(self.instances)
        cs.num_environments, cs.num_shapes = len(self.environments), len(self.shapes)
        cs.num_textures, cs.num_materials = len(self.textures), len(self.materials)
        cs.cameras, cs.instances = _ptr(self.cameras), _ptr(self.instances)
        cs.environments, cs.shapes = _ptr(self.environments), _ptr(self.shapes)
        cs.textures, cs.materials = _ptr(self.textures), _ptr(self.materials)
        for name, _, _ in self.POOLS:
            arr = getattr(self, name)
            setattr(cs, "num_" + name, len(arr))
            setattr(cs, name, _ptr(arr))
        return cs

    @staticmethod
    def from_c(cs):
        """Deep-copy a ythip_scene (e.g. exported by the reference harness)."""
        fs = FlatScene()

        def grab(ptr, n, dt):
            if not ptr or n == 0:
                return np.zeros(0, dt)
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dt, n).copy()

        fs.cameras = grab(cs.cameras, cs.num_cameras, camera_dt)
        fs.instances = grab(cs.instances, cs.num_instances, instance_dt)
        fs.environments = grab(cs.environments, cs.num_environments, environment_dt)
        fs.shapes = grab(cs.shapes, cs.num_shapes, shape_dt)
        fs.textures = grab(cs.textures, cs.num_textures, texture_dt)
        fs.materials = grab(cs.materials, cs.num_materials, material_dt)
        for name, dt, n in FlatScene.POOLS:
            cnt = getattr(cs, "num_" + name)
            setattr(fs, name, grab(getattr(cs, name), cnt * n, dt).reshape(-1, n))
        return fs


class FlatBvh:
    """scene_bvh (yocto_bvh.h:70-79) flattened: shape trees then instance tree."""

    def __init__(self, node_offset, prim_offset, nodes, primitives):
        self.node_offset = np.ascontiguousarray(node_offset, "i8")
        self.prim_offset = np.ascontiguousarray(prim_offset, "i8")
        self.nodes = np.ascontiguousarray(nodes, node_dt)
        self.primitives = np.ascontiguousarray(primitives, "i4")

    def c_struct(self):
        cb = CBvh()
        cb.num_trees = len(self.node_offset) - 1
        cb.node_offset, cb.prim_offset = _ptr(self.node_offset), _ptr(self.prim_offset)
        cb.nodes, cb.primitives = _ptr(self.nodes), _ptr(self.primitives)
        return cb

    @staticmethod
    def from_c(cb):
        nt = cb.num_trees

        def grab(ptr, n, dt):
            if not ptr or n == 0:
                return np.zeros(0, dt)
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dt, n).copy()

        no = grab(cb.node_offset, nt + 1, "i8")
        po = grab(cb.prim_offset, nt + 1, "i8")
        return FlatBvh(no, po, grab(cb.nodes, int(no[-1]), node_dt),
                       grab(cb.primitives, int(po[-1]), "i4"))

    def same_as(self, other):
        return (np.array_equal(self.node_offset, other.node_offset)
                and np.array_equal(self.prim_offset, other.prim_offset)
                and self.nodes.tobytes() == other.nodes.tobytes()
                and np.array_equal(self.primitives, other.primitives))


class FlatLights:
    """trace_lights (yocto_trace.h:126-135) flattened."""

    def __init__(self, lights, cdf):
        self.lights = np.ascontiguousarray(lights, light_dt)
        self.cdf = np.ascontiguousarray(cdf, "f4")

    def c_struct(self):
        cl = CLights()
        cl.num_lights, cl.lights = len(self.lights), _ptr(self.lights)
        cl.num_cdf, cl.cdf = len(self.cdf), _ptr(self.cdf)
        return cl

    @staticmethod
    def from_c(cl):
        def grab(ptr, n, dt):
            if not ptr or n == 0:
                return np.zeros(0, dt)
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dt, n).copy()

        return FlatLights(grab(cl.lights, cl.num_lights, light_dt),
                          grab(cl.cdf, cl.num_cdf, "f4"))


# ----------------------------------------------------------------------------
# library
# ----------------------------------------------------------------------------
class YthipError(RuntimeError):
    pass


_lib = None

_SIGNATURES = {
    "ythip_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ythip_destroy": (None, [C.c_void_p]),
    "ythip_last_error": (C.c_char_p, [C.c_void_p]),
    "ythip_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ythip_sync": (C.c_int, [C.c_void_p]),
    "ythip_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(CScene)]),
    "ythip_scene_staging": (C.c_int, [C.c_void_p, C.POINTER(CScene), C.POINTER(CScene)]),
    "ythip_upload_scene_staged": (C.c_int, [C.c_void_p]),
    "ythip_update_cameras": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "ythip_update_materials": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "ythip_update_environments": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "ythip_build_bvh": (C.c_int, [C.c_void_p, C.POINTER(CScene), C.c_int]),
    "ythip_build_own_bvh": (C.c_int, [C.c_void_p, C.POINTER(CScene)]),
    "ythip_own_bvh_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(CBuildInfo)]),
    "ythip_intersect_batch_own": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ythip_abi_version": (C.c_int, []),
    "ythip_params_size": (C.c_int, []),
    "ythip_upload_bvh": (C.c_int, [C.c_void_p, C.POINTER(CBvh)]),
    "ythip_set_bvh_builder": (C.c_int, [C.c_void_p, C.c_int, C.c_int64]),
    "ythip_bvh_build_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ythip_update_shape_vertices": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                              C.c_void_p, C.c_int64]),
    "ythip_update_instance_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "ythip_update_bvh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    "ythip_bvh_baked_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ythip_bvh_baked_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ythip_set_traversal": (C.c_int, [C.c_void_p, C.c_int]),
    "ythip_host_libm_matches": (C.c_int, [C.c_void_p]),
    "ythip_params_default": (None, [C.c_void_p]),
    "ythip_params_from_json": (C.c_int, [C.c_char_p, C.c_int64, C.c_void_p]),
    "ythip_params_to_json": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    "ythip_io_last_error": (C.c_char_p, []),
    "ythip_set_pixel_pool": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ythip_get_pixel_pool": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ythip_set_scheduler": (C.c_int, [C.c_void_p, C.c_int]),
    "ythip_get_stream_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ythip_get_scheduler": (C.c_int, [C.c_void_p]),
    "ythip_may_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ythip_set_stream_options": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "ythip_set_stream_groups": (C.c_int, [C.c_void_p, C.c_int]),
    "ythip_set_stream_finish": (C.c_int, [C.c_void_p, C.c_int]),
    "ythip_get_stream_generations": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "ythip_get_stream_walk_steps": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int32]),
    "ythip_last_launch_fastmath": (C.c_int, [C.c_void_p]),
    "ythip_ply_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "ythip_ply_read": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 9),
    "ythip_ply_close": (None, [C.c_void_p]),
    "ythip_scene_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "ythip_scene_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "ythip_scene_find_camera": (C.c_int32, [C.c_void_p, C.c_char_p]),
    "ythip_scene_name": (C.c_char_p, [C.c_void_p, C.c_int, C.c_int32]),
    "ythip_scene_close": (None, [C.c_void_p]),
    "ythip_load_scene": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p]),
    "ythip_bvh_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ythip_bvh_download": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4),
    "ythip_host_bvh_build": (C.c_int, [C.POINTER(CScene), C.c_int,
                                       C.POINTER(C.c_void_p)]),
    "ythip_host_bvh_view": (C.c_int, [C.c_void_p, C.POINTER(CBvh)]),
    "ythip_host_bvh_free": (None, [C.c_void_p]),
    "ythip_host_bvh_refit": (C.c_int, [C.c_void_p, C.POINTER(CScene), C.c_void_p, C.c_int32]),
    "ythip_host_lights_build": (C.c_int, [C.POINTER(CScene), C.POINTER(C.c_void_p)]),
    "ythip_host_lights_view": (C.c_int, [C.c_void_p, C.POINTER(CLights)]),
    "ythip_host_lights_free": (None, [C.c_void_p]),
    "ythip_build_lights": (C.c_int, [C.c_void_p, C.POINTER(CScene)]),
    "ythip_upload_lights": (C.c_int, [C.c_void_p, C.POINTER(CLights)]),
    "ythip_lights_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int64)]),
    "ythip_lights_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ythip_state_size": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]),
    "ythip_make_rngs": (C.c_int, [C.c_uint64, C.c_int64, C.c_void_p]),
    "ythip_state_create": (C.c_int, [C.c_void_p] + [C.c_int] * 4),
    "ythip_state_create_striped": (C.c_int, [C.c_void_p] + [C.c_int] * 6),
    "ythip_state_local_width": (C.c_int, [C.c_int] * 3),
    "ythip_state_upload": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int]),
    "ythip_state_download": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5
                             + [C.POINTER(C.c_int)]),
    "ythip_get_image": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ythip_get_albedo_image": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ythip_get_normal_image": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ythip_trace_sample": (C.c_int, [C.c_void_p, C.POINTER(CParams), C.c_int, C.c_int, C.c_int]),
    "ythip_tonemap_image": (C.c_int, [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ythip_denoise_default_params": (None, [C.c_void_p]),
    "ythip_denoise_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "ythip_denoise_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ythip_state_device_denoised": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ythip_state_bind_device": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5),
    "ythip_state_set_samples": (C.c_int, [C.c_void_p, C.c_int]),
    "ythip_trace_samples": (C.c_int, [C.c_void_p, C.POINTER(CParams), C.c_void_p]),
    "ythip_trace_samples_async": (C.c_int, [C.c_void_p, C.POINTER(CParams)]),
    "ythip_test_libm": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ythip_cancel": (C.c_int, [C.c_void_p]),
    "ythip_poll": (C.c_int, [C.c_void_p]),
    "ythip_state_get_samples": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "ythip_state_device_image": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ythip_create_multi": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "ythip_destroy_multi": (None, [C.c_void_p]),
    "ythip_multi_size": (C.c_int, [C.c_void_p]),
    "ythip_multi_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "ythip_multi_last_error": (C.c_char_p, [C.c_void_p]),
    "ythip_multi_state_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ythip_multi_state_upload": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int]),
    "ythip_multi_state_download": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.POINTER(C.c_int)]),
    "ythip_multi_trace_samples": (C.c_int, [C.c_void_p, C.POINTER(CParams), C.c_void_p]),
    "ythip_multi_get_image": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ythip_multi_gather_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    "ythip_intersect_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int, C.c_void_p]),
    "ythip_intersect_instance_batch": (C.c_int, [C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_int64, C.c_int,
                                                 C.c_void_p]),
    "ythip_camera_rays": (C.c_int, [C.c_void_p, C.POINTER(CParams), C.c_void_p]),
    "ythip_set_scheduling": (C.c_int, [C.c_void_p, C.c_int]),
    "ythip_set_early_miss": (C.c_int, [C.c_void_p, C.c_int]),
    "ythip_set_specialization": (C.c_int, [C.c_void_p, C.c_int]),
    "ythip_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "ythip_reset_stats": (C.c_int, [C.c_void_p]),
    "ythip_get_stats": (C.c_int, [C.c_void_p, C.POINTER(CStats)]),
}


def exported_symbols():
    return sorted(_SIGNATURES)


ABI_VERSION = 6  # include/ythip.h: YTHIP_ABI_VERSION


def load_library(path=LIB_PATH):
    """Load libythip.so and bind every symbol of include/ythip.h.  Raises if the
    library or a symbol is missing — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise YthipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ "
            f"as g; g.build()'` (hipcc --offload-arch=gfx950)")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    # the structs of this file are include/ythip.h's at ABI version 5 (a caller built against another layout of
    # ythip_params would hand the library a garbage `fastmath`)
    if lib.ythip_abi_version() != ABI_VERSION or lib.ythip_params_size() != C.sizeof(CParams):
        raise YthipError(f"{path}: ABI version {lib.ythip_abi_version()} / sizeof(ythip_params) {lib.ythip_params_size()}, "
                         f"these bindings are for version {ABI_VERSION} / {C.sizeof(CParams)} bytes")
    _lib = lib
    return lib


def state_size(camera, resolution):
    """Image size rule of make_trace_state (yocto_trace.cpp:1499-1505)."""
    lib = load_library()
    w, h = C.c_int(), C.c_int()
    cam = np.ascontiguousarray(camera, camera_dt).reshape(1)
    rc = lib.ythip_state_size(cam.ctypes.data, resolution, C.byref(w), C.byref(h))
    if rc:
        raise YthipError("ythip_state_size failed")
    return w.value, h.value


TILE = 16  # yt_kernels.h YT_TILE: width of a tile column


def slice_columns(width, col_first=0, col_stride=1):
    """Global x of every pixel column of a column-striped slice, in local order
    (ythip_state_create_striped): tile columns col_first, col_first + col_stride, ..."""
    cols = [np.arange(c * TILE, min(width, (c + 1) * TILE))
            for c in range(col_first, (width + TILE - 1) // TILE, col_stride)]
    return np.concatenate(cols) if cols else np.zeros(0, np.int64)


def make_rngs(seed, n):
    """Per-pixel PCG seeding of make_trace_state (yocto_trace.cpp:1512-1515)."""
    lib = load_library()
    out = np.zeros((n, 2), "u8")
    if lib.ythip_make_rngs(seed, n, out.ctypes.data):
        raise YthipError("ythip_make_rngs failed")
    return out


def host_make_bvh(scene, highquality=False):
    """make_scene_bvh (yocto_bvh.cpp:364-396) on the host, no GPU needed."""
    lib = load_library()
    cs = scene.c_struct()
    h = C.c_void_p()
    if lib.ythip_host_bvh_build(C.byref(cs), int(highquality), C.byref(h)):
        raise YthipError("ythip_host_bvh_build failed")
    cb = CBvh()
    lib.ythip_host_bvh_view(h, C.byref(cb))
    out = FlatBvh.from_c(cb)
    lib.ythip_host_bvh_free(h)
    return out


def host_update_bvh(scene, edited_scene, updated_shapes, highquality=False):
    """make_scene_bvh of `scene`, then update_scene_bvh (yocto_bvh.cpp:434-451) with
    `edited_scene` (same element lists, moved vertices / instances); host only."""
    lib = load_library()
    cs, ce = scene.c_struct(), edited_scene.c_struct()
    h = C.c_void_p()
    if lib.ythip_host_bvh_build(C.byref(cs), int(highquality), C.byref(h)):
        raise YthipError("ythip_host_bvh_build failed")
    vs = np.ascontiguousarray(list(updated_shapes), "i4")
    rc = lib.ythip_host_bvh_refit(h, C.byref(ce), _ptr(vs), len(vs))
    if rc:
        lib.ythip_host_bvh_free(h)
        raise YthipError("ythip_host_bvh_refit failed")
    cb = CBvh()
    lib.ythip_host_bvh_view(h, C.byref(cb))
    out = FlatBvh.from_c(cb)
    lib.ythip_host_bvh_free(h)
    return out


def host_make_lights(scene):
    """make_trace_lights (yocto_trace.cpp:1528-1581) on the host, no GPU needed."""
    lib = load_library()
    cs = scene.c_struct()
    h = C.c_void_p()
    if lib.ythip_host_lights_build(C.byref(cs), C.byref(h)):
        raise YthipError("ythip_host_lights_build failed")
    cl = CLights()
    lib.ythip_host_lights_view(h, C.byref(cl))
    out = FlatLights.from_c(cl)
    lib.ythip_host_lights_free(h)
    return out


class Context:
    """One GPU context (mirrors make_cutrace_context, yocto_cutrace.h:88)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.ythip_create(device, C.byref(h))
        if rc:
            msg = self.lib.ythip_last_error(None)
            raise YthipError(f"ythip_create({device}) failed [{rc}]: "
                             f"{msg.decode() if msg else ''}")
        self.h = h
        self.width = self.height = 0
        self.row_begin = self.row_end = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.ythip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc:
            msg = self.lib.ythip_last_error(self.h)
            raise YthipError(f"{what} failed [{rc}]: {msg.decode() if msg else ''}")

    def set_stream(self, stream_ptr):
        self._check(self.lib.ythip_set_stream(self.h, stream_ptr), "set_stream")

    def sync(self):
        self._check(self.lib.ythip_sync(self.h), "sync")

    # scene / bvh / lights -----------------------------------------------------
    def upload_scene(self, scene):
        cs = scene.c_struct()
        self._check(self.lib.ythip_upload_scene(self.h, C.byref(cs)), "upload_scene")

    def load_scene(self, path, threads=0):
        """ythip_load_scene: scene files -> pinned staging pools -> HBM; returns a FlatScene copy of
        the staged pools for make_trace_bvh / make_trace_lights / make_trace_state."""
        staged = CScene()
        rc = self.lib.ythip_load_scene(self.h, str(path).encode(), int(threads), C.byref(staged))
        if rc:
            raise YthipError(self.lib.ythip_io_last_error().decode())
        return FlatScene.from_c(staged)

    def set_pixel_pool(self, mode, workgroups=0):
        """0 never, 1 measured choice (default), 2 always — ythip_set_pixel_pool."""
        self._check(self.lib.ythip_set_pixel_pool(self.h, int(mode), int(workgroups)), "set_pixel_pool")

    def pixel_pool_info(self):
        info = CPoolInfo()
        self._check(self.lib.ythip_get_pixel_pool(self.h, C.byref(info)), "get_pixel_pool")
        return {k: getattr(info, k) for k, _ in CPoolInfo._fields_}

    def set_scheduler(self, mode):
        """0 the fused persistent kernel, 1 the streaming scheduler (csrc/yt_stream.h), 2 (default) a measured choice between the
        two per trace_state / sampler / mode / batch size — ythip_set_scheduler."""
        self._check(self.lib.ythip_set_scheduler(self.h, int(mode)), "set_scheduler")

    def set_stream_options(self, order=-1, cell_bits=-1, phased=-1):
        """The sort of the streaming scheduler (ythip_set_stream_options; -1 keeps): order 0 octant major / 1 cell major /
        2 unsorted, cell_bits 1..5, phased 0 / 1."""
        self._check(self.lib.ythip_set_stream_options(self.h, int(order), int(cell_bits), int(phased)), "set_stream_options")

    def set_stream_groups(self, groups):
        """1..8 chains of generations side by side (default 2) — ythip_set_stream_groups."""
        self._check(self.lib.ythip_set_stream_groups(self.h, int(groups)), "set_stream_groups")

    def set_stream_finish(self, permille):
        """A group's queue at `permille` thousandths of its path slots goes to the tail kernel (default 250; 0 never,
        1000 the whole batch) — ythip_set_stream_finish."""
        self._check(self.lib.ythip_set_stream_finish(self.h, int(permille)), "set_stream_finish")

    def stream_walk_steps(self, generation, fetch=False):
        """Profiling: choose the generation whose per-ray walk lengths the next profiled streamed batch logs (fetch=False), or
        fetch them (int32 per path slot, queue order per group) — ythip_get_stream_walk_steps."""
        if not fetch:
            self._check(self.lib.ythip_get_stream_walk_steps(self.h, int(generation), None, 0), "get_stream_walk_steps")
            return None
        out = np.zeros(self.stream_info()["path_slots"], "i4")
        self._check(self.lib.ythip_get_stream_walk_steps(self.h, int(generation), _ptr(out), len(out)), "get_stream_walk_steps")
        return out

    def stream_generations(self):
        """Queue length of every generation of the last streamed batch (profiling mode 1 during the batch)."""
        out, n = np.zeros(8192, "i4"), C.c_int32(0)
        self._check(self.lib.ythip_get_stream_generations(self.h, _ptr(out), len(out), C.byref(n)), "get_stream_generations")
        return out[:n.value]

    def stream_info(self):
        info = CStreamInfo()
        self._check(self.lib.ythip_get_stream_info(self.h, C.byref(info)), "get_stream_info")
        return {k: getattr(info, k) for k, _ in CStreamInfo._fields_}

    def last_launch_fastmath(self):
        """The mode the last trace launch ran: 0 bit-exact, 1 the tolerance-mode kernels (yt_fast.hip), 2 the own-tree
        kernels (yt_owntree.hip) — params.fastmath where such a kernel exists."""
        return int(self.lib.ythip_last_launch_fastmath(self.h))

    def update_cameras(self, cameras):
        """Re-upload only the cameras (interactive camera edits, apps/ytrace.cpp:189-204)."""
        cams = np.ascontiguousarray(cameras, camera_dt)
        self._check(self.lib.ythip_update_cameras(self.h, cams.ctypes.data, len(cams)),
                    "update_cameras")

    def update_materials(self, materials):
        """In-place material edits (same count as the resident scene)."""
        m = np.ascontiguousarray(materials, material_dt)
        self._check(self.lib.ythip_update_materials(self.h, m.ctypes.data, len(m)), "update_materials")

    def update_environments(self, environments):
        e = np.ascontiguousarray(environments, environment_dt)
        self._check(self.lib.ythip_update_environments(self.h, e.ctypes.data, len(e)), "update_environments")

    def make_trace_bvh(self, scene, highquality=False):
        cs = scene.c_struct()
        self._check(self.lib.ythip_build_bvh(self.h, C.byref(cs), int(highquality)),
                    "build_bvh")

    def make_own_bvh(self, scene=None):
        """The own tree of trace_params(fastmath=2): built next to the resident reference tree (make_trace_bvh first).
        Without a scene: from the geometry as resident (it follows update_shape_vertices / update_instance_frames)."""
        cs = scene.c_struct() if scene is not None else None
        self._check(self.lib.ythip_build_own_bvh(self.h, C.byref(cs) if cs is not None else None), "build_own_bvh")

    def own_bvh_info(self):
        n, l4, info = C.c_int64(), C.c_int64(), CBuildInfo()
        self._check(self.lib.ythip_own_bvh_info(self.h, C.byref(n), C.byref(l4), C.byref(info)), "own_bvh_info")
        d = {k: getattr(info, k) for k, _ in CBuildInfo._fields_}
        d.update(num_nodes=n.value, num_leaf4=l4.value, bytes=64 * n.value + 16 * l4.value)
        return d

    def set_bvh_builder(self, mode, min_prims=0):
        """mode "device" (default: shapes >= min_prims are built on the GPU) or "host"."""
        self._check(self.lib.ythip_set_bvh_builder(self.h, {"host": 0, "device": 1}[mode], min_prims),
                    "set_bvh_builder")

    def update_shape_vertices(self, shape, positions=None, normals=None, radius=None):
        """New positions / normals / radii of a resident shape (same counts; element lists stay)."""
        p = None if positions is None else np.ascontiguousarray(positions, "f4").reshape(-1, 3)
        nrm = None if normals is None else np.ascontiguousarray(normals, "f4").reshape(-1, 3)
        r = None if radius is None else np.ascontiguousarray(radius, "f4").reshape(-1)
        cnt = lambda a: 0 if a is None else len(a)  # noqa: E731
        self._check(self.lib.ythip_update_shape_vertices(
            self.h, int(shape), _ptr(p), cnt(p), _ptr(nrm), cnt(nrm), _ptr(r), cnt(r)), "update_shape_vertices")

    def update_instance_frames(self, instances, frames):
        ids = np.ascontiguousarray(instances, "i4").reshape(-1)
        fr = np.ascontiguousarray(frames, "f4").reshape(len(ids), 12)
        self._check(self.lib.ythip_update_instance_frames(self.h, _ptr(ids), len(ids), _ptr(fr)),
                    "update_instance_frames")

    def update_bvh(self, updated_instances=(), updated_shapes=()):
        """update_scene_bvh (yocto_bvh.cpp:434-451): refit, topology kept."""
        vi = np.ascontiguousarray(list(updated_instances), "i4")
        vs = np.ascontiguousarray(list(updated_shapes), "i4")
        self._check(self.lib.ythip_update_bvh(self.h, _ptr(vi), len(vi), _ptr(vs), len(vs)), "update_bvh")

    def bvh_build_info(self):
        info = CBuildInfo()
        self._check(self.lib.ythip_bvh_build_info(self.h, C.byref(info)), "bvh_build_info")
        return {k: getattr(info, k) for k, _ in CBuildInfo._fields_}

    def bvh_baked_sizes(self):
        """Bytes of the baked traversal data: pair records (64 B) + quad records (128 B) +
        leaf data (16 B per float4)."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self.lib.ythip_bvh_baked_sizes(self.h, C.byref(a), C.byref(b)), "bvh_baked_sizes")
        return int(a.value) * (64 + 128) + int(b.value) * 16

    def download_baked_bvh(self):
        """The traversal layout: pairs [n, 16] f4, leafdata [m, 4] f4."""
        n, m = C.c_int64(), C.c_int64()
        self._check(self.lib.ythip_bvh_baked_sizes(self.h, C.byref(n), C.byref(m)), "bvh_baked_sizes")
        pairs, leaf = np.zeros((n.value, 16), "f4"), np.zeros((m.value, 4), "f4")
        quads = np.zeros((n.value, 32), "f4")
        self._check(self.lib.ythip_bvh_baked_download(self.h, _ptr(pairs), _ptr(leaf), _ptr(quads)),
                    "bvh_baked_download")
        return pairs, leaf, quads

    def upload_bvh(self, bvh):
        cb = bvh.c_struct()
        self._check(self.lib.ythip_upload_bvh(self.h, C.byref(cb)), "upload_bvh")

    def download_bvh(self):
        nt, nn, npr = C.c_int32(), C.c_int64(), C.c_int64()
        self._check(self.lib.ythip_bvh_sizes(self.h, C.byref(nt), C.byref(nn),
                                             C.byref(npr)), "bvh_sizes")
        no = np.zeros(nt.value + 1, "i8")
        po = np.zeros(nt.value + 1, "i8")
        nodes = np.zeros(nn.value, node_dt)
        prims = np.zeros(npr.value, "i4")
        self._check(self.lib.ythip_bvh_download(
            self.h, no.ctypes.data, po.ctypes.data, _ptr(nodes), _ptr(prims)),
            "bvh_download")
        return FlatBvh(no, po, nodes, prims)

    def make_trace_lights(self, scene):
        cs = scene.c_struct()
        self._check(self.lib.ythip_build_lights(self.h, C.byref(cs)), "build_lights")

    def upload_lights(self, lights):
        cl = lights.c_struct()
        self._check(self.lib.ythip_upload_lights(self.h, C.byref(cl)), "upload_lights")

    def download_lights(self):
        nl, nc = C.c_int32(), C.c_int64()
        self._check(self.lib.ythip_lights_sizes(self.h, C.byref(nl), C.byref(nc)),
                    "lights_sizes")
        lights = np.zeros(nl.value, light_dt)
        cdf = np.zeros(nc.value, "f4")
        self._check(self.lib.ythip_lights_download(self.h, _ptr(lights), _ptr(cdf)),
                    "lights_download")
        return FlatLights(lights, cdf)

    # state --------------------------------------------------------------------
    def make_trace_state(self, scene, params, rows=None, rngs=None, cols=None):
        """make_trace_state (yocto_trace.cpp:1495-1520) for rows [r0, r1) and,
        with cols=(first, stride), the 16-pixel tile columns first, first +
        stride, ... of the frame (ythip_state_create_striped)."""
        w, h = state_size(scene.cameras[params.camera], params.resolution)
        r0, r1 = (0, h) if rows is None else rows
        c0, cs = (0, 1) if cols is None else cols
        self._check(self.lib.ythip_state_create_striped(self.h, w, h, r0, r1, c0, cs),
                    "state_create_striped")
        self.width, self.height, self.row_begin, self.row_end = w, h, r0, r1
        self.local_width = self.lib.ythip_state_local_width(w, c0, cs)
        if rngs is None:
            rngs = make_rngs(params.seed, w * h)
        sl = np.ascontiguousarray(
            np.asarray(rngs).reshape(h, w, 2)[r0:r1, slice_columns(w, c0, cs)].reshape(-1, 2))
        self._check(self.lib.ythip_state_upload(self.h, None, None, None, None,
                                                sl.ctypes.data, 0), "state_upload")
        return w, h

    @property
    def npixels(self):
        return self.local_width * (self.row_end - self.row_begin)

    def upload_state(self, image=None, albedo=None, normal=None, hits=None,
                     rngs=None, samples=0):
        arrs = [None if a is None else np.ascontiguousarray(a, dt)
                for a, dt in [(image, "f4"), (albedo, "f4"), (normal, "f4"),
                              (hits, "i4"), (rngs, "u8")]]
        self._check(self.lib.ythip_state_upload(
            self.h, *[None if a is None else a.ctypes.data for a in arrs], samples),
            "state_upload")

    def download_state(self):
        n = self.npixels
        image = np.zeros((n, 4), "f4")
        albedo = np.zeros((n, 3), "f4")
        normal = np.zeros((n, 3), "f4")
        hits = np.zeros(n, "i4")
        rngs = np.zeros((n, 2), "u8")
        samples = C.c_int()
        self._check(self.lib.ythip_state_download(
            self.h, image.ctypes.data, albedo.ctypes.data, normal.ctypes.data,
            hits.ctypes.data, rngs.ctypes.data, C.byref(samples)), "state_download")
        return dict(image=image, albedo=albedo, normal=normal, hits=hits, rngs=rngs,
                    samples=samples.value, width=self.local_width,
                    height=self.row_end - self.row_begin)

    def get_image(self):
        """get_image (yocto_trace.cpp:1694-1708): width x rows x 4 linear floats."""
        h = self.row_end - self.row_begin
        image = np.zeros((h, self.local_width, 4), "f4")
        self._check(self.lib.ythip_get_image(self.h, image.ctypes.data), "get_image")
        return image

    def get_albedo_image(self):
        """get_albedo_image (yocto_trace.cpp:1769-1780): {albedo.xyz, 1} per pixel."""
        image = np.zeros((self.row_end - self.row_begin, self.local_width, 4), "f4")
        self._check(self.lib.ythip_get_albedo_image(self.h, image.ctypes.data), "get_albedo_image")
        return image

    def get_normal_image(self):
        """get_normal_image (yocto_trace.cpp:1781-1791): {normal.xyz, 1} per pixel."""
        image = np.zeros((self.row_end - self.row_begin, self.local_width, 4), "f4")
        self._check(self.lib.ythip_get_normal_image(self.h, image.ctypes.data), "get_normal_image")
        return image

    def tonemap_image(self, exposure=0.0, filmic=False, srgb=True):
        """tonemap_image on the device: (float [h, w, 4], bytes [h, w, 4])."""
        h = self.row_end - self.row_begin
        ldr = np.zeros((h, self.local_width, 4), "f4")
        ldrb = np.zeros((h, self.local_width, 4), "u1")
        self._check(self.lib.ythip_tonemap_image(self.h, exposure, int(filmic), int(srgb),
                                                 ldr.ctypes.data, ldrb.ctypes.data), "tonemap_image")
        return ldr, ldrb

    # denoiser (csrc/yt_denoise.h; the slot of denoise_image, yocto_trace.cpp:1794-1872) -------
    def denoise_params(self, **kw):
        p = CDenoiseParams()
        self.lib.ythip_denoise_default_params(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise TypeError(f"unknown denoise parameter {k}")
            setattr(p, k, v)
        return p

    def denoise_image(self, render, albedo, normal, **kw):
        """Host images (render [h, w, 4], albedo / normal [h, w, 3], float32) -> denoised [h, w, 4]."""
        render = np.ascontiguousarray(render, "f4")
        albedo = np.ascontiguousarray(albedo, "f4")
        normal = np.ascontiguousarray(normal, "f4")
        h, w = render.shape[:2]
        if render.shape != (h, w, 4) or albedo.shape != (h, w, 3) or normal.shape != (h, w, 3):
            raise ValueError("denoise_image: render [h,w,4], albedo [h,w,3], normal [h,w,3]")
        out = np.zeros((h, w, 4), "f4")
        p = self.denoise_params(**kw)
        self._check(self.lib.ythip_denoise_image(self.h, C.byref(p), w, h, render.ctypes.data, albedo.ctypes.data,
                                                 normal.ctypes.data, out.ctypes.data), "denoise_image")
        return out

    def denoise_state(self, download=True, **kw):
        """The resident whole-frame state's image, filtered on the device; [h, w, 4] or None."""
        out = np.zeros((self.height, self.width, 4), "f4") if download else None
        p = self.denoise_params(**kw)
        self._check(self.lib.ythip_denoise_state(self.h, C.byref(p), out.ctypes.data if download else None),
                    "denoise_state")
        return out

    def bind_device_state(self, image, albedo, normal, hits, rngs):
        self._check(self.lib.ythip_state_bind_device(self.h, image, albedo, normal,
                                                     hits, rngs), "state_bind_device")

    # hot path -------------------------------------------------------------------
    def trace_samples(self, params, stop=None):
        self._check(self.lib.ythip_trace_samples(
            self.h, C.byref(params), None if stop is None else stop.ctypes.data),
            "trace_samples")

    def trace_sample(self, params, i, j, sample):
        """trace_sample (yocto_trace.cpp:1461-1492): one sample of frame pixel (i, j)."""
        self._check(self.lib.ythip_trace_sample(self.h, C.byref(params), i, j, sample), "trace_sample")

    LIBM_FUNCTIONS = ["sinf", "cosf", "expf", "exp2f", "logf", "atanf", "acosf", "atan2f", "powf", "fmodf", "sqrtf",
                      "div"]

    def test_libm(self, fn, x, y=None):
        """yt_libm.h's restatement of glibc's `fn` evaluated on the device (test entry)."""
        x = np.ascontiguousarray(x, np.float32)
        y = None if y is None else np.ascontiguousarray(y, np.float32)
        out = np.zeros_like(x)
        self._check(self.lib.ythip_test_libm(self.h, self.LIBM_FUNCTIONS.index(fn), x.ctypes.data,
                                             None if y is None else y.ctypes.data, x.size, out.ctypes.data),
                    "test_libm")
        return out

    def cancel(self):
        self._check(self.lib.ythip_cancel(self.h), "cancel")

    def trace_samples_async(self, params):
        self._check(self.lib.ythip_trace_samples_async(self.h, C.byref(params)),
                    "trace_samples_async")

    def intersect_batch(self, rays, find_any=False):
        rays = np.ascontiguousarray(rays, ray_dt)
        hits = np.zeros(len(rays), hit_dt)
        self._check(self.lib.ythip_intersect_batch(
            self.h, _ptr(rays), len(rays), int(find_any), _ptr(hits)),
            "intersect_batch")
        return hits

    def intersect_batch_own(self, rays, instances=None):
        """The same batch through the own tree's walk (the traversal of fastmath = 2): a measuring entry."""
        rays = np.ascontiguousarray(rays, ray_dt)
        inst = None if instances is None else np.ascontiguousarray(instances, "i4")
        hits = np.zeros(len(rays), hit_dt)
        self._check(self.lib.ythip_intersect_batch_own(self.h, _ptr(inst), _ptr(rays), len(rays), _ptr(hits)),
                    "intersect_batch_own")
        return hits

    def intersect_instance_batch(self, instances, rays, find_any=False):
        rays = np.ascontiguousarray(rays, ray_dt)
        instances = np.ascontiguousarray(instances, "i4")
        hits = np.zeros(len(rays), hit_dt)
        self._check(self.lib.ythip_intersect_instance_batch(
            self.h, _ptr(instances), _ptr(rays), len(rays), int(find_any),
            _ptr(hits)), "intersect_instance_batch")
        return hits

    def camera_rays(self, params):
        rays = np.zeros(self.npixels, ray_dt)
        self._check(self.lib.ythip_camera_rays(self.h, C.byref(params), _ptr(rays)),
                    "camera_rays")
        return rays

    # measurement ------------------------------------------------------------------
    def set_specialization(self, enable):
        self._check(self.lib.ythip_set_specialization(self.h, int(enable)), "set_specialization")

    def set_scheduling(self, adaptive_wait):
        self._check(self.lib.ythip_set_scheduling(self.h, int(adaptive_wait)), "set_scheduling")

    def set_early_miss(self, enable):
        self._check(self.lib.ythip_set_early_miss(self.h, int(enable)), "set_early_miss")

    def host_libm_matches(self):
        """Does the host's libm agree with the device's restatement of glibc 2.35 (ythip_host_libm_matches)?"""
        rc = self.lib.ythip_host_libm_matches(self.h)
        if rc < 0:
            self._check(-rc, "host_libm_matches")
        return rc == 1

    def set_traversal(self, mode):
        """"binary" | "wide" | "auto" (default): which BVH walk the kernels use."""
        self._check(self.lib.ythip_set_traversal(self.h, {"binary": 0, "wide": 1, "auto": 2}[mode]),
                    "set_traversal")

    def set_profiling(self, mode):
        self._check(self.lib.ythip_set_profiling(self.h, mode), "set_profiling")

    def reset_stats(self):
        self._check(self.lib.ythip_reset_stats(self.h), "reset_stats")

    def get_stats(self):
        s = CStats()
        self._check(self.lib.ythip_get_stats(self.h, C.byref(s)), "get_stats")
        return {k: getattr(s, k) for k, _ in CStats._fields_}


def algorithmic_bytes(stats):
    """bytes = 32 N_node + 52 N_tri + 68 N_quad + 44 N_line
    + 24 N_point + 60 N_inst + 248 N_shade + 120 N_sample (reference layouts)."""
    return (32 * stats["nodes"] + 52 * stats["triangles"] + 68 * stats["quads"]
            + 44 * stats["lines"] + 24 * stats["points"] + 60 * stats["instances"]
            + 248 * stats["shades"] + 120 * stats["samples"])


def traversal_bytes(stats):
    """The traversal kernel's share of the formula above (no shading/state)."""
    return (32 * stats["nodes"] + 52 * stats["triangles"] + 68 * stats["quads"]
            + 44 * stats["lines"] + 24 * stats["points"] + 60 * stats["instances"])
