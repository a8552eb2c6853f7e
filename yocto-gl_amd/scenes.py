"""Synthetic scene generators (host side, numpy float32) for the BASELINE.json
configs and the parity tests.  They restate the few reference generators the
configs are defined with, in float32 with the reference's operation order, so
that e.g. the 1M-triangle plane is bit-identical to
`make_recty({1000,500},{10,10})` + `quads_to_triangles` (checked in
tests/test_host.py against the compiled reference).

  make_quads / make_rect / make_recty   libs/yocto/yocto_shape.cpp:546-627
  quads_to_triangles                    libs/yocto/yocto_shape.cpp:2535-2543
  make_uvsphere                         libs/yocto/yocto_shape.cpp:783-796
  lookat_frame                          libs/yocto/yocto_math.h:2348-2358
"""
import numpy as np

from ythip import FlatScene, IDENTITY_FRAME

f32 = np.float32


def _normalize(v):
    v = np.asarray(v, f32)
    l = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2], dtype=f32)
    return (v / l).astype(f32) if l != 0 else v


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2],
                     a[0] * b[1] - a[1] * b[0]], f32)


def lookat_frame(eye, center, up=(0, 1, 0)):
    eye, center, up = (np.asarray(x, f32) for x in (eye, center, up))
    w = _normalize(eye - center)
    u = _normalize(_cross(up, w))
    v = _normalize(_cross(w, u))
    return np.concatenate([u, v, w, eye]).astype(f32)


def make_quads(steps, scale, uvscale=(1, 1)):
    sx, sy = int(steps[0]), int(steps[1])
    i = np.arange(sx + 1, dtype=f32)
    j = np.arange(sy + 1, dtype=f32)
    u = (i / f32(sx)).astype(f32)
    v = (j / f32(sy)).astype(f32)
    uu, vv = np.meshgrid(u, v)  # [sy+1, sx+1], index j*(sx+1)+i
    px = ((f32(2) * uu - f32(1)) * f32(scale[0])).astype(f32)
    py = ((f32(2) * vv - f32(1)) * f32(scale[1])).astype(f32)
    positions = np.stack([px, py, np.zeros_like(px)], -1).reshape(-1, 3)
    normals = np.tile(np.array([0, 0, 1], f32), (len(positions), 1))
    texcoords = np.stack([uu * f32(uvscale[0]), (f32(1) - vv) * f32(uvscale[1])],
                         -1).reshape(-1, 2).astype(f32)
    ii, jj = np.meshgrid(np.arange(sx, dtype=np.int32), np.arange(sy, dtype=np.int32))
    a = jj * (sx + 1) + ii
    quads = np.stack([a, a + 1, a + (sx + 1) + 1, a + (sx + 1)], -1).reshape(-1, 4)
    return dict(positions=positions, normals=normals, texcoords=texcoords,
                quads=quads.astype(np.int32))


def make_rect(steps, scale, uvscale=(1, 1)):
    return make_quads(steps, scale, uvscale)


def make_recty(steps, scale, uvscale=(1, 1)):
    s = make_quads(steps, scale, uvscale)
    p, n = s["positions"], s["normals"]
    s["positions"] = np.stack([p[:, 0], p[:, 2], -p[:, 1]], -1).astype(f32)
    s["normals"] = np.stack([n[:, 0], n[:, 2], n[:, 1]], -1).astype(f32)
    return s


def quads_to_triangles(quads):
    q = np.asarray(quads, np.int32)
    t1 = q[:, [0, 1, 3]]
    t2 = q[:, [2, 3, 1]]
    keep = q[:, 2] != q[:, 3]
    out = np.empty((len(q) * 2, 3), np.int32)
    out[0::2] = t1
    out[1::2] = t2
    mask = np.ones(len(q) * 2, bool)
    mask[1::2] = keep
    return out[mask]


def triangulated(shape):
    s = dict(shape)
    s["triangles"] = quads_to_triangles(s.pop("quads"))
    return s


def make_uvsphere(steps, scale):
    s = make_quads(steps, (1, 1))
    uv = s["texcoords"]
    ax = (f32(2) * f32(np.pi) * uv[:, 0]).astype(f32)
    ay = (f32(np.pi) * (f32(1) - uv[:, 1])).astype(f32)
    p = np.stack([np.cos(ax) * np.sin(ay), np.sin(ax) * np.sin(ay), np.cos(ay)],
                 -1).astype(f32) * f32(scale)
    n = p / np.linalg.norm(p, axis=1, keepdims=True)
    s["positions"], s["normals"] = p.astype(f32), n.astype(f32)
    return s


def add_shape(scene, shape):
    return scene.add_shape(shape["positions"], triangles=shape.get("triangles"),
                           quads=shape.get("quads"), lines=shape.get("lines"),
                           points=shape.get("points"), normals=shape.get("normals"),
                           texcoords=shape.get("texcoords"), colors=shape.get("colors"),
                           radius=shape.get("radius"))


# ----------------------------------------------------------------------------
# BASELINE.json configs (SURVEY.md §8d recipes)
# ----------------------------------------------------------------------------
def plane_scene(steps=(1000, 500), scale=(10, 10)):
    """cfg2: make_recty(steps, scale) triangulated (1,000,000 triangles at the
    default steps), matte 0.7, identity instance, constant white environment,
    16:9 camera at (0,3,8) looking at the origin."""
    sc = FlatScene()
    eye, center = (0, 3, 8), (0, 0, 0)
    focus = float(np.sqrt(f32(0 * 0 + 3 * 3 + 8 * 8), dtype=f32))
    sc.add_camera(lookat_frame(eye, center), lens=0.035, film=0.036, aspect=16 / 9,
                  focus=focus, aperture=0.0)
    sh = add_shape(sc, triangulated(make_recty(steps, scale)))
    m = sc.add_material("matte", color=(0.7, 0.7, 0.7))
    sc.add_instance(sh, m)
    sc.add_environment((1, 1, 1))
    return sc


def instanced_scene(grid=100, sphere_steps=(32, 16), radius=0.04):
    """cfg4: grid x grid instances of one triangulated uv-sphere (1024 triangles
    at 32x16) on a 0.1-spaced lattice, same environment / camera as cfg2."""
    sc = FlatScene()
    focus = float(np.sqrt(f32(73), dtype=f32))
    sc.add_camera(lookat_frame((0, 3, 8), (0, 0, 0)), lens=0.035, film=0.036,
                  aspect=16 / 9, focus=focus, aperture=0.0)
    sh = add_shape(sc, triangulated(make_uvsphere(sphere_steps, radius)))
    m = sc.add_material("matte", color=(0.7, 0.7, 0.7))
    k = np.arange(grid * grid)
    frames = np.tile(IDENTITY_FRAME, (grid * grid, 1))
    frames[:, 9] = ((k % grid - grid // 2) * 0.1).astype(f32)
    frames[:, 10] = f32(radius)
    frames[:, 11] = ((k // grid - grid // 2) * 0.1).astype(f32)
    sc.add_instances(frames, sh, m)
    sc.add_environment((1, 1, 1))
    return sc


# ----------------------------------------------------------------------------
# flat scene <-> npz (scene fixtures: tests/golden/*.npz)
# ----------------------------------------------------------------------------
SCENE_FIELDS = ["cameras", "instances", "environments", "shapes", "textures", "materials"] + \
    [p[0] for p in FlatScene.POOLS]


def save_scene(path, sc):
    np.savez_compressed(path, **{k: getattr(sc, k) for k in SCENE_FIELDS})


def load_scene(path):
    z = np.load(path)
    sc = FlatScene()
    for k in SCENE_FIELDS:
        setattr(sc, k, z[k])
    return sc


def load_corpus_scene(name, root=None):
    """A scene of the reference's own test corpus (tests/_version43: features1, materials1-4, ...) from the committed
    fixtures tests/golden/scenes/<name>.json + blobs/ — the flat pools the reference's load_scene + tesselate_subdivs
    produced, stored by content hash (tests/golden/make_scene_fixtures.py).  Textured, mixed-material scenes: what the
    general kernel class renders."""
    import json
    import os
    if root is None:
        root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
    with open(os.path.join(root, name + ".json")) as f:
        manifest = json.load(f)
    sc = FlatScene()
    for field, key in manifest.items():
        setattr(sc, field, np.load(os.path.join(root, "blobs", key + ".npz"))["a"])
    return sc


def cornell_1m_scene(base, n=316):
    """cfg2b (the north star's "1M-triangle Cornell-box-style scene"):
    the reference's Cornell box `base` (make_cornellbox, yocto_scene.cpp:970-1075, as
    exported to tests/golden/cornellbox.npz) with each of the five walls replaced by an
    n x n grid of quads over the same corners, triangulated — 5 * 2 * 316^2 + 26 =
    998,586 triangles; boxes, area light, materials and camera unchanged.  Deep paths
    (every ray hits) and area-light pdf walks."""
    sc = FlatScene()
    sc.cameras = base.cameras.copy()
    sc.materials = base.materials.copy()
    i = np.arange(n + 1, dtype=f32) / f32(n)
    uu, vv = np.meshgrid(i, i)  # index j * (n + 1) + i
    ii, jj = np.meshgrid(np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32))
    a = jj * (n + 1) + ii
    quads = np.stack([a, a + 1, a + (n + 1) + 1, a + (n + 1)], -1).reshape(-1, 4)
    tris = quads_to_triangles(quads)
    for k in range(len(base.shapes)):
        arr = base.shape_arrays(k)
        if k < 5:  # a wall: bilinear grid over its corners p0 p1 p2 p3
            p0, p1, p2, p3 = (arr["positions"][c] for c in range(4))
            u, v = uu[..., None], vv[..., None]
            pos = ((p0 * (f32(1) - u) + p1 * u) * (f32(1) - v) + (p3 * (f32(1) - u) + p2 * u) * v)
            sc.add_shape(pos.reshape(-1, 3).astype(f32), triangles=tris)
        else:
            sc.add_shape(arr["positions"], triangles=arr["triangles"])
    for inst in base.instances:
        sc.add_instance(int(inst["shape"]), int(inst["material"]), inst["frame"])
    return sc


def _pcg32_floats(seed, seq, n):
    """rand1f(rng) x n of make_rng(seed, seq) — yocto_sampling.h:187-232 (PCG32, integer-exact)."""
    M = (1 << 64) - 1
    inc = ((seq << 1) | 1) & M
    state = 0

    def advance():
        nonlocal state
        old = state
        state = (old * 6364136223846793005 + inc) & M
        xs = (((old >> 18) ^ old) >> 27) & 0xffffffff
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xffffffff

    advance()
    state = (state + seed) & M
    advance()
    bits = np.fromiter(((advance() >> 9) | 0x3f800000 for _ in range(n)), np.uint32, n)
    return bits.view(np.float32) - f32(1)


HAIR_ROOTS = None  # tests/golden/hair_roots.npz (set by the first call)


def hair_scene(roots_file=None, steps=8, length=(0.2, 0.2), radius=(0.002, 0.001), seed=7):
    """BASELINE configs[4] exactly: make_hair(make_sphere(32, 1), {8, 100000}, {0.2, 0.2},
    {0.002, 0.001}) — 800,000 line segments on 900,000 vertices with radii — over its base sphere, subsurface hair
    material, matte base, constant environment, the camera of the parity tests.  The strand roots / normals are the
    g++ reference's sample_shape output (the one compiler-dependent part, stored in tests/golden/hair_roots.npz by
    tests/golden/make_hair_fixture.py); the rest of make_hair (yocto_shape.cpp:1279-1305 with noise = clump = 0) is
    restated here in float32 with the reference's operation order and checked against the reference's arrays byte for
    byte (tests/test_host.py), so bench.py renders the scene the parity tests render without touching oracle/."""
    import os
    global HAIR_ROOTS
    if roots_file is None:
        roots_file = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hair_roots.npz")
    if HAIR_ROOTS is None or HAIR_ROOTS[0] != roots_file:
        z = np.load(roots_file)
        HAIR_ROOTS = (roots_file, z["roots"].astype(f32), z["normals"].astype(f32),
                      {k[5:]: z[k] for k in z.files if k.startswith("base_")})  # make_sphere(32, 1) as the reference built it
    bpos, bnorm = HAIR_ROOTS[1], HAIR_ROOTS[2]
    base = HAIR_ROOTS[3]
    strands = len(bpos)
    sc = FlatScene()
    # the camera of the parity tests (SURVEY.md §8d names none for this scene: tests/test_gpu_baseline_configs.py::hair_scene)
    sc.add_camera(lookat_frame((0, 0.5, 3.2), (0, 0, 0)), lens=0.035, film=0.036, aspect=16 / 9,
                  focus=float(np.sqrt(f32(0.5 * 0.5 + 3.2 * 3.2), dtype=f32)), aperture=0.0)
    s_base = add_shape(sc, base)
    # blen = lerp(len.x, len.y, rand1f(rng)) with rng = make_rng(seed, 3)
    r = _pcg32_floats(seed, 3, strands)
    blen = (f32(length[0]) * (f32(1) - r) + f32(length[1]) * r).astype(f32)
    # make_lines(steps, ...): u = i / steps.x per vertex of a strand; positions = bpos + bnorm * u * blen
    u = (np.arange(steps + 1, dtype=f32) / f32(steps)).astype(f32)
    pos = (bpos[:, None, :] + (bnorm[:, None, :] * u[None, :, None]) * blen[:, None, None]).astype(f32).reshape(-1, 3)
    nrm = np.repeat(bnorm, steps + 1, 0)
    rad = np.tile((f32(radius[0]) * (f32(1) - u) + f32(radius[1]) * u).astype(f32), strands)
    v = (np.arange(steps + 1, dtype=f32) / f32(steps))  # make_lines texcoords: {i / steps.x, j / steps.y}
    w = (np.arange(strands, dtype=f32) / f32(strands - 1)) if strands > 1 else np.zeros(1, f32)
    tex = np.stack([np.tile(v, strands), np.repeat(w, steps + 1)], -1).astype(f32)
    k = (np.arange(strands, dtype=np.int32) * (steps + 1))[:, None] + np.arange(steps, dtype=np.int32)[None, :]
    lines = np.stack([k, k + 1], -1).reshape(-1, 2).astype(np.int32)
    s_hair = sc.add_shape(pos, lines=lines, normals=nrm, texcoords=tex, radius=rad)
    m_base = sc.add_material(type="matte", color=(0.7, 0.7, 0.7))
    m_hair = sc.add_material(type="subsurface", color=(0.8, 0.6, 0.4), roughness=0.3, scattering=(0.5, 0.5, 0.5))
    sc.add_instance(s_base, m_base)
    sc.add_instance(s_hair, m_hair)
    sc.add_environment((1, 1, 1))
    return sc
