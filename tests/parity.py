"""Shared helpers for the parity tests: seeded test scenes (numpy-authored flat
scenes fed IDENTICALLY to the HIP path and to the oracle), oracle access, and
comparison utilities.  Test infrastructure only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ythip as yt  # noqa: E402
import scenes as ysc  # noqa: E402
import refyocto as ry  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
f32 = np.float32


def have_ref():
    return ry.available()


# ----------------------------------------------------------------------------
# flat scene <-> npz
# ----------------------------------------------------------------------------
save_scene, load_scene, _FIELDS = ysc.save_scene, ysc.load_scene, ysc.SCENE_FIELDS


# ----------------------------------------------------------------------------
# test scenes
# ----------------------------------------------------------------------------
def scene_cornellbox():
    """make_cornellbox() (yocto_scene.cpp:970-1075) exported from the reference."""
    return load_scene(os.path.join(GOLDEN, "cornellbox.npz"))


def scene_plane():
    return ysc.plane_scene(steps=(100, 50))


def _smooth_texture(rng, h, w, lo=0.0, hi=1.0):
    base = rng.random((h // 4 + 2, w // 4 + 2, 4)).astype(f32)
    img = np.kron(base, np.ones((4, 4, 1), f32))[:h, :w]
    return (lo + (hi - lo) * img).astype(f32)


def scene_materials():
    """One sphere per material type (rough and delta variants), a textured floor,
    an emissive quad, a lat-long environment texture, an opacity material and a
    normal map: exercises every BSDF lobe, textures, area+environment lights."""
    rng = np.random.default_rng(7)
    sc = yt.FlatScene()
    sc.add_camera(ysc.lookat_frame((0, 1.6, 5.5), (0, 0.6, 0)), lens=0.05, film=0.036,
                  aspect=2.0, focus=5.6, aperture=0.02)
    # textures
    t_floor = sc.add_texture((_smooth_texture(rng, 32, 32) * 255).astype(np.uint8))  # sRGB bytes
    t_rough = sc.add_texture(_smooth_texture(rng, 16, 16, 0.2, 1.0), linear=True)
    t_env = sc.add_texture(_smooth_texture(rng, 16, 32, 0.1, 1.5), linear=True)
    nm = _smooth_texture(rng, 16, 16, 0.35, 0.65)
    nm[..., 2] = 1.0
    t_normal = sc.add_texture(nm, linear=True)
    t_alpha = _smooth_texture(rng, 16, 16)
    t_alpha[..., 3] = (t_alpha[..., 0] > 0.5).astype(f32)
    t_alpha = sc.add_texture(t_alpha, linear=True, nearest=True)
    # floor (quads, textured)
    floor = ysc.make_recty((4, 4), (4, 4), (2, 2))
    s_floor = ysc.add_shape(sc, floor)
    m_floor = sc.add_material("matte", color=(0.8, 0.8, 0.8), color_tex=t_floor)
    sc.add_instance(s_floor, m_floor)
    # spheres: quads with normals + texcoords
    sph_q = ysc.make_uvsphere((16, 8), 0.35)
    sph_t = ysc.triangulated(ysc.make_uvsphere((16, 8), 0.35))
    s_sq = ysc.add_shape(sc, sph_q)
    s_st = ysc.add_shape(sc, sph_t)
    mats = [
        dict(type="matte", color=(0.7, 0.3, 0.3)),
        dict(type="glossy", color=(0.3, 0.7, 0.3), roughness=0.3),
        dict(type="reflective", color=(0.8, 0.7, 0.4), roughness=0.25),
        dict(type="reflective", color=(0.8, 0.8, 0.8), roughness=0.0),
        dict(type="transparent", color=(0.9, 0.8, 0.7), roughness=0.2),
        dict(type="transparent", color=(0.9, 0.9, 0.9), roughness=0.0),
        dict(type="refractive", color=(0.9, 0.95, 0.9), roughness=0.15, trdepth=0.5),
        dict(type="refractive", color=(0.95, 0.9, 0.9), roughness=0.0, trdepth=0.5),
        dict(type="subsurface", color=(0.8, 0.6, 0.4), roughness=0.3,
             scattering=(0.5, 0.5, 0.5), trdepth=0.2),
        dict(type="volumetric", color=(0.6, 0.7, 0.9), scattering=(0.6, 0.6, 0.6),
             scanisotropy=0.3, trdepth=0.3),
        dict(type="gltfpbr", color=(0.6, 0.5, 0.8), roughness=0.4, metallic=0.6,
             roughness_tex=t_rough),
        dict(type="matte", color=(0.8, 0.8, 0.2), opacity=0.6),
        dict(type="glossy", color=(0.5, 0.5, 0.5), roughness=0.2, normal_tex=t_normal),
        dict(type="matte", color=(0.9, 0.9, 0.9), color_tex=t_alpha),
    ]
    for k, m in enumerate(mats):
        mid = sc.add_material(**m)
        fr = yt.IDENTITY_FRAME.copy()
        col, row = k % 7, k // 7
        fr[9:12] = [(col - 3) * 0.85, 0.36 + row * 0.9, -row * 0.8]
        if k % 3 == 1:  # a non-rigid (scaled) frame
            fr[0], fr[4], fr[8] = 1.2, 0.8, 1.1
        sc.add_instance(s_sq if k % 2 == 0 else s_st, mid, fr)
    # area light: emissive quad above
    light = ysc.make_recty((1, 1), (0.8, 0.8))
    light["positions"] = light["positions"] * f32([1, -1, 1])  # face down
    s_light = ysc.add_shape(sc, dict(positions=light["positions"], quads=light["quads"]))
    m_light = sc.add_material("matte", emission=(12, 11, 9))
    fr = yt.IDENTITY_FRAME.copy()
    fr[9:12] = [0, 3.0, 0.5]
    sc.add_instance(s_light, m_light, fr)
    # environment with a texture, rotated frame
    sc.add_environment((0.6, 0.7, 0.9), emission_tex=t_env,
                       frame=ysc.lookat_frame((0, 0, 0), (0.3, 0.1, -1)))
    return sc


def scene_instances():
    sc = ysc.instanced_scene(grid=12, sphere_steps=(8, 4), radius=0.04)
    # make every third instance non-rigid
    fr = sc.instances["frame"]
    fr[::3, 0] = 1.5
    fr[::3, 4] = 0.6
    fr[1::3, 8] = 1.25
    sc.instances["frame"] = fr
    # a second material, glossy, for half the instances
    m2 = sc.add_material("glossy", color=(0.3, 0.5, 0.8), roughness=0.2)
    sc.instances["material"][::2] = m2
    return sc


def scene_lines_points():
    """Hair-like line segments (radius, tangents as normals) + a point cloud
    (radius), subsurface/matte materials, constant environment."""
    rng = np.random.default_rng(11)
    sc = yt.FlatScene()
    sc.add_camera(ysc.lookat_frame((0, 0.4, 3.0), (0, 0.2, 0)), lens=0.05, film=0.036,
                  aspect=1.5, focus=3.0, aperture=0.0)
    nstrands, nseg = 300, 6
    roots = (rng.random((nstrands, 3)).astype(f32) - f32(0.5)) * f32([1.6, 0.2, 1.0])
    dirs = rng.normal(size=(nstrands, 3)).astype(f32) * f32(0.3) + f32([0, 1, 0])
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    t = np.linspace(0, 0.8, nseg + 1, dtype=f32)
    pos = roots[:, None, :] + dirs[:, None, :] * t[None, :, None] + \
        f32(0.05) * np.sin(f32(8) * t)[None, :, None] * rng.normal(size=(nstrands, 1, 3)).astype(f32)
    pos = pos.reshape(-1, 3).astype(f32)
    base = (np.arange(nstrands) * (nseg + 1))[:, None] + np.arange(nseg)[None, :]
    lines = np.stack([base, base + 1], -1).reshape(-1, 2).astype(np.int32)
    tang = np.repeat(dirs, nseg + 1, axis=0).astype(f32)
    radius = np.tile(np.linspace(0.012, 0.004, nseg + 1, dtype=f32), nstrands)
    s_hair = sc.add_shape(pos, lines=lines, normals=tang, radius=radius,
                          texcoords=rng.random((len(pos), 2)).astype(f32))
    m_hair = sc.add_material("subsurface", color=(0.8, 0.6, 0.4), roughness=0.3,
                             scattering=(0.5, 0.5, 0.5))
    sc.add_instance(s_hair, m_hair)
    npts = 400
    ppos = (rng.random((npts, 3)).astype(f32) - f32(0.5)) * f32([2.0, 1.0, 1.0]) + f32([0, 0.5, -0.8])
    s_pts = sc.add_shape(ppos, points=np.arange(npts, dtype=np.int32),
                         radius=np.full(npts, 0.03, f32),
                         colors=rng.random((npts, 4)).astype(f32) * f32(0.5) + f32(0.5))
    m_pts = sc.add_material("matte", color=(0.9, 0.9, 0.9))
    fr = yt.IDENTITY_FRAME.copy()
    fr[0], fr[4], fr[8] = 1.1, 0.9, 1.0
    sc.add_instance(s_pts, m_pts, fr)
    floor = ysc.triangulated(ysc.make_recty((2, 2), (2, 2)))
    s_floor = ysc.add_shape(sc, floor)
    sc.add_instance(s_floor, sc.add_material("matte", color=(0.5, 0.5, 0.5)))
    sc.add_environment((1, 1, 1))
    return sc


def scene_cornell_1m(n=316):
    """cfg2b: the Cornell box with 1M-triangle walls (scenes.cornell_1m_scene)."""
    return ysc.cornell_1m_scene(scene_cornellbox(), n)


REF_SCENE_DIR = os.path.join(GOLDEN, "scenes")


def ref_scene_names():
    """Scenes of the reference's own test corpus (tests/_version43) available as
    fixtures — see tests/golden/make_scene_fixtures.py."""
    return sorted(f[:-5] for f in os.listdir(REF_SCENE_DIR) if f.endswith(".json")) \
        if os.path.isdir(REF_SCENE_DIR) else []


def load_ref_scene(name):
    """A scene of the reference's test corpus as the reference's own loader read it
    (flat POD layout; arrays stored by content hash)."""
    return ysc.load_corpus_scene(name, REF_SCENE_DIR)


SCENES = {
    "cornellbox": scene_cornellbox,
    "plane": scene_plane,
    "materials": scene_materials,
    "instances": scene_instances,
    "lines_points": scene_lines_points,
}


# ----------------------------------------------------------------------------
# oracle / GPU plumbing
# ----------------------------------------------------------------------------
class RefBundle:
    """scene + make_trace_bvh + make_trace_lights inside the compiled reference."""

    def __init__(self, flat, highquality=False):
        self.scene = ry.RefScene.from_flat(flat)
        self.bvh = ry.RefBvh(self.scene, highquality)
        self.lights = ry.RefLights(self.scene)

    def render(self, params, state_init=None):
        st = ry.RefState(self.scene, params)
        if state_init is not None:
            st.set(**state_init)
        while st.samples < params.samples:
            ry.trace_samples(st, self.scene, self.bvh, self.lights, params)
        return st.get()


REF_RENDERS = {}  # (name, resolution, spp) -> the compiled reference's trace_state: full-size renders cost its CPU seconds once per session


def ref_render_cached(key, flat, params):
    if key not in REF_RENDERS:
        REF_RENDERS[key] = RefBundle(flat).render(params)
    return REF_RENDERS[key]


def gpu_context(flat, highquality=False, device=0):
    ctx = yt.Context(device)
    ctx.upload_scene(flat)
    ctx.make_trace_bvh(flat, highquality)
    ctx.make_trace_lights(flat)
    return ctx


def gpu_render(ctx, flat, params, rows=None, rngs=None, cols=None):
    ctx.make_trace_state(flat, params, rows=rows, rngs=rngs, cols=cols)
    n = 0
    while n < params.samples:
        ctx.trace_samples(params)
        n += params.batch
    return ctx.download_state()


def render_pair(name, sampler="path", resolution=64, samples=4, batch=None, **kw):
    flat = SCENES[name]()
    params = yt.trace_params(sampler=sampler, resolution=resolution, samples=samples,
                             batch=batch or samples, **kw)
    ctx = gpu_context(flat)
    gpu = gpu_render(ctx, flat, params)
    ctx.close()
    ref = RefBundle(flat).render(params) if have_ref() else None
    return dict(gpu=gpu, ref=ref, flat=flat, params=params)


# ----------------------------------------------------------------------------
# scene edits that keep the element lists (update_scene_bvh, yocto_bvh.cpp:434-451)
# ----------------------------------------------------------------------------
def edit_scene(flat, seed=11, shapes=None, amount=0.05):
    """Deep copy of `flat` with the vertices of `shapes` (default: every other
    shape, at least one) displaced, their normals perturbed, their radii rescaled, a few coordinates snapped
    to +0 / -0 (the merge order decides the sign of such a box face), and every
    third instance re-posed.  Returns (edited, moved_shapes, moved_instances)."""
    import copy
    rng = np.random.default_rng(seed)
    ed = copy.deepcopy(flat)
    if shapes is None:
        shapes = list(range(0, len(flat.shapes), 2))
    for s in shapes:
        off, n = int(ed.shapes[s]["positions_offset"]), int(ed.shapes[s]["num_positions"])
        if n == 0:
            continue
        pos = ed.positions[off:off + n]
        ext = max(float(np.abs(pos).max()), 1e-3)
        pos += (rng.normal(size=pos.shape) * amount * ext).astype(f32)
        k = max(n // 16, 1)
        pos[rng.integers(0, n, k), rng.integers(0, 3, k)] = f32(0.0)
        pos[rng.integers(0, n, k), rng.integers(0, 3, k)] = f32(-0.0)
        noff, nn = int(ed.shapes[s]["normals_offset"]), int(ed.shapes[s]["num_normals"])
        if nn > 0:  # shading normals move with the surface
            nrm = ed.normals[noff:noff + nn] + (rng.normal(size=(nn, 3)) * 0.1).astype(f32)
            ed.normals[noff:noff + nn] = (nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-6)).astype(f32)
        roff, rn = int(ed.shapes[s]["radius_offset"]), int(ed.shapes[s]["num_radius"])
        if rn > 0:
            ed.radius[roff:roff + rn] *= (0.5 + rng.random((rn, 1))).astype(f32)
    moved = list(range(0, len(flat.instances), 3))
    for i in moved:
        fr = ed.instances[i]["frame"].reshape(4, 3).copy()
        a = rng.random() * 0.5
        rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], f32)
        fr[:3] = (fr[:3] @ rot).astype(f32)
        fr[3] += (rng.normal(size=3) * amount).astype(f32)
        ed.instances[i]["frame"] = fr.reshape(12)
    return ed, list(shapes), moved


def apply_edit_to_ref(ref_scene, edited, shapes, instances):
    for s in shapes:
        a = edited.shape_arrays(s)
        ref_scene.set_vertices(s, a["positions"], a["normals"], a["radius"])
    for i in instances:
        ref_scene.set_instance_frame(i, edited.instances[i]["frame"])


def apply_edit_to_gpu(ctx, edited, shapes, instances):
    for s in shapes:
        a = edited.shape_arrays(s)
        ctx.update_shape_vertices(s, a["positions"], a["normals"], a["radius"])
    if instances:
        ctx.update_instance_frames(instances, np.stack([edited.instances[i]["frame"] for i in instances]))


# ----------------------------------------------------------------------------
# comparison
# ----------------------------------------------------------------------------
STATE_KEYS = ["image", "albedo", "normal", "hits", "rngs"]


def assert_identical(a, b, what=""):
    """The whole trace_state, bit for bit.  Since round 2 the device evaluates the reference
    platform's libm (yt_libm.h), so every float of the path is the reference's and there is no
    tolerance left to state: image, albedo, normal, hits and the rng streams are equal bytes."""
    for k in STATE_KEYS:
        if a[k].tobytes() != b[k].tobytes():
            ne = a[k].reshape(len(a[k]), -1) != b[k].reshape(len(b[k]), -1)
            raise AssertionError(f"{what}: {k} differs in {int(ne.any(1).sum())} of {len(a[k])} pixels")


def image_stats(a, b):
    """Per-pixel relative differences between two radiance images (n,4)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b).max(axis=1)
    scale = np.maximum(np.abs(b).max(axis=1), 1e-3)
    rel = d / scale
    return dict(frac_exact=float((a == b).all(axis=1).mean()),
                frac_1e4=float((rel <= 1e-4).mean()),
                frac_1e2=float((rel <= 1e-2).mean()),
                mean_rel=float(abs(a.mean() - b.mean()) / max(abs(b.mean()), 1e-6)),
                max_rel=float(rel.max()))


def random_rays(flat, n, seed=3):
    """Seeded rays from inside the scene's bounds towards random directions, plus
    axis-aligned and degenerate directions (zeros → 1/0 = inf in intersect_bbox)."""
    rng = np.random.default_rng(seed)
    lo, hi = flat.positions.min(0) - 0.5, flat.positions.max(0) + 0.5
    rays = np.zeros(n, yt.ray_dt)
    rays["o"] = (lo + (hi - lo) * rng.random((n, 3))).astype(f32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["d"] = d.astype(f32)
    k = max(n // 16, 1)
    axes = np.eye(3, dtype=f32)
    rays["d"][:k] = axes[rng.integers(0, 3, k)] * rng.choice([-1, 1], (k, 1)).astype(f32)
    rays["d"][k:2 * k, 1] = 0  # one zero component
    rays["tmin"] = f32(1e-4)
    rays["tmax"] = np.finfo(f32).max
    rays["tmax"][2 * k:3 * k] = (rng.random(k) * 3).astype(f32)  # finite tmax
    return rays


def hits_equal(a, b):
    """Bit-exact equality of scene_intersection records; on a miss only the hit
    flag is compared (the reference leaves the other fields at their defaults)."""
    a, b = np.asarray(a), np.asarray(b)
    if not np.array_equal(a["hit"], b["hit"]):
        return False
    m = a["hit"] != 0
    return all(a[f][m].tobytes() == b[f][m].tobytes()
               for f in ["instance", "element", "u", "v", "distance"])
