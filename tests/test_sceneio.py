"""Scene FILES into the flat pools on the host (`-m "not gpu"`): ythip_scene_open /
ythip_scene_read (yocto-gl_amd/csrc/yt_sceneio.hip) against the reference's own load_scene, run live from
oracle/_ref — every pool of the flattened scene byte for byte:

  * the reference's test corpus where it is present (this container; not the GPU box);
  * scenes the reference's own save_scene writes (JSON + PLY + PNG / HDR), anywhere oracle/_ref travelled;
  * hand-written scene.json files for the loader's fix-ups (lookat, missing camera, missing radius, defaults,
    number classes) and its refusals;
  * PNG files of every colour type / bit depth / interlacing / tRNS / filter, and Radiance HDR files in the
    run-length, flat and "not run-length after all" layouts, written by the small encoders below and read by
    the reference through stb_image.
"""
import json
import os
import struct
import zlib

import numpy as np
import pytest

import parity as P
from parity import ry, yt

needs_ref = pytest.mark.skipif(not P.have_ref(), reason="oracle/_ref not built / did not travel")
RECORDS = ["cameras", "instances", "environments", "shapes", "textures", "materials"]
POOLS = [p[0] for p in yt.FlatScene.POOLS]
CORPUS = os.path.join(os.environ.get("YOCTO_REF", "/root/reference"), "tests", "_version43")
CORPUS40 = os.path.join(os.environ.get("YOCTO_REF", "/root/reference"), "tests", "_version40")
CORPUS_TOP = os.path.join(os.environ.get("YOCTO_REF", "/root/reference"), "tests")  # the scenes the reference's test script renders today


def assert_same_scene(got, ref, what):
    for k in RECORDS + POOLS:
        a, b = np.ascontiguousarray(getattr(got, k)), np.ascontiguousarray(getattr(ref, k))
        assert a.shape[0] == b.shape[0], f"{what}: {k} count {a.shape[0]} != {b.shape[0]}"
        assert a.tobytes() == b.tobytes(), f"{what}: {k}"


def both(path):
    got, names, camera = yt.load_scene_file(path)
    ref = ry.RefScene.load(path).flat()
    assert_same_scene(got, ref, os.path.basename(str(path)))
    return got, names, camera


# ---------------------------------------------------------------------------------------------------
@needs_ref
@pytest.mark.skipif(not os.path.isdir(CORPUS), reason="the reference's test corpus is not present (GPU box)")
@pytest.mark.parametrize("corpus", [CORPUS, CORPUS40, CORPUS_TOP], ids=["format-4.2", "format-4.0", "current"])
def test_the_reference_corpus(corpus):
    seen = 0
    for name in sorted(os.listdir(corpus)):
        path = f"{corpus}/{name}/{name}.json"
        if not os.path.exists(path):
            continue
        if json.load(open(path)).get("subdivs"):  # (an empty "subdivs" group is nothing to tesselate)
            with pytest.raises(yt.YthipError, match="subdivs"):
                yt.load_scene_file(path)
            continue
        try:
            ref = ry.RefScene.load(path).flat()
        except RuntimeError as e:
            # shapes3 names three files that are not there: both loaders must say so, in the same words.  WHICH of the
            # three the reference names depends on its threads; this reader always names the first in file order.
            with pytest.raises(yt.YthipError) as mine:
                yt.load_scene_file(path)
            head = f"cannot load {path} since cannot open "
            assert str(e).startswith(head) and str(mine.value).startswith(head)
            assert not os.path.exists(str(e)[len(head):]) and not os.path.exists(str(mine.value)[len(head):])
            continue
        got, names, _ = yt.load_scene_file(path)
        assert_same_scene(got, ref, name)
        if corpus == CORPUS:
            assert names["shapes"] == [s["name"] for s in json.load(open(path))["shapes"]]
        seen += 1
    assert seen >= (5 if corpus == CORPUS_TOP else 13)


@needs_ref
@pytest.mark.parametrize("name", ["cornellbox", "materials", "instances", "lines_points"])
def test_scenes_written_by_the_reference(name, tmp_path):
    """save_scene's output (format 4.2 JSON, binary PLY, PNG / HDR textures) read back both ways; one thread
    and many give the same pools."""
    src = ry.RefScene.from_flat(P.SCENES[name]())
    path = str(tmp_path / "scene.json")
    src.save(path)
    got, names, camera = both(path)
    one, _, _ = yt.load_scene_file(path, threads=1)
    assert_same_scene(one, got, name + " (1 thread)")
    assert len(names["instances"]) == len(got.instances) and camera == 0


# ---------------------------------------------------------------------------------------------------
# hand-written scene.json: fix-ups, defaults, number classes
# ---------------------------------------------------------------------------------------------------
def write_ply(path, positions, triangles=None, lines=None, points=None, radius=None):
    n = len(positions)
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {n}", "property float x", "property float y",
            "property float z"]
    if radius is not None:
        head.append("property float radius")
    cols = [np.asarray(positions, "<f4")]
    if radius is not None:
        cols.append(np.asarray(radius, "<f4").reshape(n, 1))
    body = np.concatenate(cols, 1).tobytes()
    for kind, elems in (("face", triangles), ("line", lines), ("point", points)):
        if elems is None:
            continue
        head += [f"element {kind} {len(elems)}", "property list uchar int vertex_indices"]
        for e in elems:
            body += struct.pack("<B", len(e)) + np.asarray(e, "<i4").tobytes()
    open(path, "wb").write(("\n".join(head) + "\nend_header\n").encode() + body)


@pytest.fixture
def scene_dir(tmp_path):
    os.mkdir(tmp_path / "shapes")
    rng = np.random.default_rng(5)
    write_ply(tmp_path / "shapes/tri.ply", rng.uniform(-1, 1, (5, 3)), triangles=[[0, 1, 2], [2, 3, 4]])
    write_ply(tmp_path / "shapes/hair.ply", rng.uniform(-2, 3, (4, 3)), lines=[[0, 1], [1, 2, 3]])
    write_ply(tmp_path / "shapes/dots.ply", rng.uniform(-1, 1, (3, 3)), points=[[0], [1, 2]], radius=[0.1, 0.2, 0.3])
    return tmp_path


def write_scene(d, doc):
    path = str(d / "scene.json")
    json.dump(doc, open(path, "w"))
    return path


BASE = {"asset": {"version": "4.2", "copyright": "nobody"},
        "shapes": [{"name": "tri", "uri": "shapes/tri.ply"}, {"name": "hair", "uri": "shapes/hair.ply"},
                   {"name": "dots", "uri": "shapes/dots.ply"}]}


@needs_ref
def test_fixups_defaults_and_number_classes(scene_dir):
    doc = dict(BASE)
    doc["cameras"] = [
        {"name": "plain"},  # all defaults
        {"name": "look", "lookat": [1, 2, 3, 0, 0.5, 0, 0, 1, 0], "lens": 1, "aspect": 2, "orthographic": True},
        {"name": "framed", "frame": [1, 0, 0, 0, 1, 0, 0, 0, 1, 0.25, -3, 7.5, 99], "film": 0.024, "aperture": 1e-3, "focus": 12},
    ]
    doc["materials"] = [
        {"name": "m0"},
        {"name": "m1", "type": "glossy", "color": [1, 0.5, 0.25], "roughness": 0.2, "emission": [0, 0, 0], "color_tex": -1},
        {"name": "m2", "type": "no such type", "ior": 2, "trdepth": 0.5, "scattering": [0.1, 0.2, 0.3], "scanisotropy": -0.5,
         "opacity": 0.75, "metallic": 1, "normal_tex": 3, "emission_tex": 0, "roughness_tex": 1, "scattering_tex": 2},
        {"name": "m3", "type": "gltfpbr", "color": [0.0, -0.0, 1e-40, 5]},
        {"name": "m4", "type": "volumetric", "ior": True},
    ]
    doc["instances"] = [
        {"name": "i0", "shape": 0, "material": 1},
        {"name": "i1", "shape": 1, "material": 2, "frame": [0, 1, 0, -1, 0, 0, 0, 0, 1, 5, 6, 7]},
        {"name": "i2", "shape": 2, "material": 0, "lookat": [3, 3, 3, 0, 0, 0, 0, 0, 1]},
        {"name": "i3"},
    ]
    doc["environments"] = [{"name": "e0", "emission": [1, 2, 3]},
                           {"name": "e1", "lookat": [0, 0, 0, 1, 1, 0, 0, 1, 0], "emission_tex": -1}]
    got, names, camera = both(write_scene(scene_dir, doc))
    assert got.shapes["num_radius"].tolist() == [0, 4, 3]  # add_missing_radius on the lines, not on the points that have one
    assert np.all(got.radius[:4] == np.float32(0.001))
    assert names["cameras"] == ["plain", "look", "framed"] and names["materials"][2] == "m2" and camera == 0
    assert got.materials["type"].tolist() == [0, 1, 0, 7, 6]


@needs_ref
def test_a_scene_without_cameras_gets_the_reference_default(scene_dir):
    doc = dict(BASE)
    doc["instances"] = [{"shape": 0, "frame": [1, 0, 0, 0, 1, 0, 0, 0, 1, 2, 0, 0]},
                        {"shape": 1, "frame": [0, 0, 1, 0, 1, 0, -1, 0, 0, 0, 5, -4]}, {"shape": 2}]
    got, names, camera = both(write_scene(scene_dir, doc))
    assert len(got.cameras) == 1 and names["cameras"] == ["camera"] and camera == 0


def test_find_camera_follows_the_reference_order(scene_dir):
    lib = yt.load_library()
    import ctypes as C
    doc = dict(BASE)
    doc["cameras"] = [{"name": "a"}, {"name": "camera1"}, {"name": "camera"}, {"name": "b"}]
    h, counts = C.c_void_p(), yt.CScene()
    assert lib.ythip_scene_open(write_scene(scene_dir, doc).encode(), C.byref(h), C.byref(counts)) == 0
    assert lib.ythip_scene_find_camera(h, b"b") == 3
    assert lib.ythip_scene_find_camera(h, b"zzz") == 2  # "default" absent -> "camera"
    assert lib.ythip_scene_find_camera(h, None) == 2
    assert lib.ythip_scene_name(h, 0, 1) == b"camera1" and lib.ythip_scene_name(h, 3, 2) == b"dots"
    assert lib.ythip_scene_name(h, 0, 4) is None and lib.ythip_scene_name(h, 9, 0) is None
    lib.ythip_scene_close(h)


@pytest.mark.parametrize("mutate, message", [
    (lambda d: d["asset"].__setitem__("version", "4.1"), "format 4.1"),
    (lambda d: d["asset"].__setitem__("version", "3.0"), "cannot parse"),
    (lambda d: d.__setitem__("subdivs", [{"name": "s"}]), "subdivs"),
    (lambda d: d["shapes"].append({"uri": "shapes/none.ply"}), "cannot open"),
    (lambda d: d["shapes"].append({"uri": "shapes/thing.obj"}), "unsupported format"),
    (lambda d: d.__setitem__("textures", [{"uri": "t.gif"}]), "unsupported format"),
    (lambda d: d.__setitem__("cameras", [{"lens": "wide"}]), "cannot parse"),
    (lambda d: d.__setitem__("cameras", [{"frame": [1, 2, 3]}]), "cannot parse"),
    (lambda d: d.__setitem__("materials", [{"name": 5}]), "cannot parse"),
    (lambda d: d.__setitem__("textures", [{"uri": "x.png", "linear": 1}]), "cannot parse"),
])
def test_refusals_are_loud(scene_dir, mutate, message):
    doc = json.loads(json.dumps(BASE))
    mutate(doc)
    with pytest.raises(yt.YthipError, match=message):
        yt.load_scene_file(write_scene(scene_dir, doc))


def test_broken_json_and_wrong_extension(scene_dir):
    path = str(scene_dir / "scene.json")
    open(path, "w").write('{"asset": {"version": "4.2"}, "cameras": [}')
    with pytest.raises(yt.YthipError, match="cannot parse"):
        yt.load_scene_file(path)
    with pytest.raises(yt.YthipError, match="unsupported format"):
        yt.load_scene_file(str(scene_dir / "scene.obj"))
    with pytest.raises(yt.YthipError, match="cannot open"):
        yt.load_scene_file(str(scene_dir / "absent.json"))


@needs_ref
def test_the_same_refusals_as_the_reference_where_it_refuses_too(scene_dir):
    """Where the reference's loader fails on these inputs, the messages are the same text."""
    for mutate in (lambda d: d["shapes"].append({"uri": "shapes/none.ply"}), lambda d: d.__setitem__("cameras", [{"lens": "wide"}]),
                   lambda d: d["asset"].__setitem__("version", "3.0")):
        doc = json.loads(json.dumps(BASE))
        mutate(doc)
        path = write_scene(scene_dir, doc)
        with pytest.raises(RuntimeError) as theirs:
            ry.RefScene.load(path)
        with pytest.raises(yt.YthipError) as mine:
            yt.load_scene_file(path)
        assert str(mine.value) == str(theirs.value)


@needs_ref
def test_json_corner_cases_behave_as_the_reference_json_library(scene_dir):
    """Number classes, containers of the wrong kind, escapes, a BOM: the same pools or the same refusal."""
    base = '{"asset":{"version":"4.2"},"shapes":[{"uri":"shapes/tri.ply"}],%s}'
    cases = ['"cameras":[{"lens":1e400}]', '"cameras":[{"lens":-0}]', '"cameras":[{"lens":-0.0}]', '"cameras":[{"lens":12345678901234567890}]',
             '"cameras":[{"lens":123456789012345678901234567890}]', '"cameras":[{"lens":null}]', '"cameras":null', '"cameras":{}',
             '"cameras":{"b":{"lens":2},"a":{"lens":3}}', '"cameras":5', '"cameras":[{"lens":1,"lens":2}]',
             '"cameras":[{"name":"\\u00e9\\ud83d\\ude00\\n"}]', '"cameras":[{"orthographic":1}]', '"cameras":[{"lens":1.5e-50}]',
             '"cameras":[{"lens":01}]', '"cameras":[{"lens":1.}]', '"cameras":[{"lens":.5}]', '"cameras":[{"lens":+1}]', '"cameras":[{"lens":1}],',
             '"materials":[{"type":5}]', '"materials":[{"type":null}]', '"materials":[{"color_tex":1.9}]', '"materials":[{"color_tex":-1.9}]',
             '"materials":[{"color_tex":true}]', '"instances":[{"shape":0,"material":3000000000}]', '"instances":[{"shape":0,"lookat":[1,2,3]}]',
             '"textures":[]', '"subdivs":[]', '"subdivs":null', '"cameras":[5]', '"cameras":[[]]', '"cameras":[null]', '"cameras":["x"]',
             '"environments":[{"emission":[1,2]}]', '"environments":[{"emission":{"a":1}}]', '"environments":{"z":{"emission":[1,2,3]}}']
    texts = [base % c for c in cases] + ['\ufeff' + base % '"cameras":[{"lens":2}]', base % '"cameras":[{"lens":2}]' + " x", "", "{"]
    path = str(scene_dir / "scene.json")
    outcomes = set()
    for text in texts:
        open(path, "w", encoding="utf-8").write(text)
        try:
            ref, theirs = ry.RefScene.load(path).flat(), None
        except RuntimeError as e:
            ref, theirs = None, str(e)
        try:
            got, mine = yt.load_scene_file(path)[0], None
        except yt.YthipError as e:
            got, mine = None, str(e)
        assert mine == theirs, text
        if ref is not None:
            assert_same_scene(got, ref, text)
        outcomes.add(theirs is None)
    assert outcomes == {True, False}


# ---------------------------------------------------------------------------------------------------
# formats 4.0 / 4.1: groups of named elements, references by name
# ---------------------------------------------------------------------------------------------------
def old_scene(tmp_path, rng):
    os.mkdir(tmp_path / "shapes"), os.mkdir(tmp_path / "textures")
    write_ply(tmp_path / "shapes/floor.ply", rng.uniform(-1, 1, (4, 3)), triangles=[[0, 1, 2], [2, 3, 0]])
    write_ply(tmp_path / "shapes/hair.ply", rng.uniform(-1, 1, (5, 3)), lines=[[0, 1, 2], [3, 4]])
    write_png(tmp_path / "textures/wood.png", rng.integers(0, 256, (5, 6, 3)), 2, 8)
    write_png(tmp_path / "textures/bump.png", rng.integers(0, 256, (4, 4, 1)), 0, 8)
    rgbe = rng.integers(1, 255, (3, 9, 4))
    write_hdr(tmp_path / "textures/sky.hdr", rgbe, "rle")
    write_hdr(tmp_path / "textures/both.hdr", rgbe, "flat")  # both.hdr AND both.png exist: .hdr is first in find_path's list
    write_png(tmp_path / "textures/both.png", rng.integers(0, 256, (2, 2, 4)), 6, 8)
    return {
        "asset": {"copyright": "x"},
        "cameras": {"main": {"lookat": [0, 1, 5, 0, 0, 0, 0, 1, 0], "ortho": True, "lens": 0.1},
                    "second": {"frame": [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 3], "orthographic": False, "ortho": False}},
        "environments": {"sky": {"emission": [1, 1, 1], "emission_tex": "sky", "lookat": [0, 0, 0, 1, 0, 0, 0, 1, 0]}},
        "materials": {"floor": {"type": "metallic", "color": [0.5, 0.4, 0.3], "color_tex": "wood", "normal_tex": "bump"},
                      "fog": {"type": "volume", "scattering": [0.1, 0.2, 0.3], "emission_tex": "both", "roughness_tex": "wood"},
                      "odd": {"type": "reflective"}},
        "instances": {"a": {"shape": "hair", "material": "fog", "lookat": [1, 2, 3, 0, 0, 0, 0, 0, 1]},
                      "b": {"shape": "floor", "material": "floor"}},
        "objects": {"c": {"shape": "hair", "material": "odd", "frame": [0, 1, 0, -1, 0, 0, 0, 0, 1, 1, 2, 3]}, "d": {}},
    }


@needs_ref
def test_old_format_references_by_name(tmp_path):
    doc = old_scene(tmp_path, np.random.default_rng(4))
    got, names, _ = both(write_scene(tmp_path, doc))
    assert names["textures"] == ["sky", "wood", "bump", "both"] and names["shapes"] == ["hair", "floor"]  # order of first mention
    assert got.materials["type"].tolist() == [2, 6, 0] and got.textures["is_float"].tolist() == [1, 0, 0, 1]
    assert got.cameras["orthographic"].tolist() == [1, 0] and names["instances"] == ["a", "b", "c", "d"]


@needs_ref
def test_old_format_refusals_match_the_reference(tmp_path):
    base = old_scene(tmp_path, np.random.default_rng(5))
    for mutate in (lambda d: d["instances"]["a"].__setitem__("material", "nobody"),   # "missing key" -> cannot parse
                   lambda d: d["instances"]["a"].__setitem__("shape", "nowhere"),     # shapes/nowhere.ply does not exist
                   lambda d: d["materials"]["floor"].__setitem__("color_tex", "none"),  # textures/none.hdr (first of the list)
                   lambda d: d["cameras"].__setitem__("bad", 5),
                   lambda d: d["instances"]["a"].__setitem__("shape", 7)):
        doc = json.loads(json.dumps(base))
        mutate(doc)
        path = write_scene(tmp_path, doc)
        with pytest.raises(RuntimeError) as theirs:
            ry.RefScene.load(path)
        with pytest.raises(yt.YthipError) as mine:
            yt.load_scene_file(path)
        assert str(mine.value) == str(theirs.value)
    doc = json.loads(json.dumps(base))
    doc["objects"]["c"]["instance"] = "grid"
    with pytest.raises(yt.YthipError, match="instanced from a PLY file"):
        yt.load_scene_file(write_scene(tmp_path, doc))
    doc = json.loads(json.dumps(base))
    doc["subdivs"] = {"s": {"shape": "floor"}}
    with pytest.raises(yt.YthipError, match="subdivs"):
        yt.load_scene_file(write_scene(tmp_path, doc))


# ---------------------------------------------------------------------------------------------------
# PNG
# ---------------------------------------------------------------------------------------------------
def _chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))


def _pack_rows(samples, depth):
    """samples [h][w*channels] of `depth`-bit values -> packed big-endian rows."""
    h, n = samples.shape
    if depth == 8:
        return [samples[y].astype("u1").tobytes() for y in range(h)]
    if depth == 16:
        return [samples[y].astype(">u2").tobytes() for y in range(h)]
    per = 8 // depth
    rows = []
    for y in range(h):
        row = np.zeros((n + per - 1) // per * per, "u1")
        row[:n] = samples[y]
        row = row.reshape(-1, per)
        acc = np.zeros(len(row), "u1")
        for k in range(per):
            acc |= (row[:, k] << ((per - 1 - k) * depth)).astype("u1")
        rows.append(acc.tobytes())
    return rows


def _filter_rows(rows, bpp, rng):
    out, prior = b"", None
    for row in rows:
        cur = np.frombuffer(row, "u1").astype(int)
        pri = np.zeros_like(cur) if prior is None else prior
        left = np.concatenate([np.zeros(bpp, int), cur[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
        upleft = np.concatenate([np.zeros(bpp, int), pri[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
        kind = int(rng.integers(0, 5))
        if kind == 0:
            enc = cur
        elif kind == 1:
            enc = cur - left
        elif kind == 2:
            enc = cur - pri
        elif kind == 3:
            enc = cur - (left + pri) // 2
        else:
            p = left + pri - upleft
            pa, pb, pc = abs(p - left), abs(p - pri), abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, pri, upleft))
            enc = cur - pred
        out += bytes([kind]) + (enc % 256).astype("u1").tobytes()
        prior = cur
    return out


def write_png(path, samples, color, depth, interlace=False, palette=None, trns=None, rng=None, extra_chunks=()):
    """samples [h][w][channels] integer sample values of `depth` bits."""
    rng = rng or np.random.default_rng(0)
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    raw = b""
    if not interlace:
        raw = _filter_rows(_pack_rows(samples.reshape(h, w * ch), depth), bpp, rng)
    else:
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += _filter_rows(_pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, rng)
    data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, int(interlace)))
    for kind, body in extra_chunks:
        data += _chunk(kind, body)
    if palette is not None:
        data += _chunk(b"PLTE", np.asarray(palette, "u1").tobytes())
    if trns is not None:
        data += _chunk(b"tRNS", trns)
    z = zlib.compress(raw, 6)
    cut = max(1, len(z) // 3)  # several IDAT chunks
    for k in range(0, len(z), cut):
        data += _chunk(b"IDAT", z[k:k + cut])
    open(path, "wb").write(data + _chunk(b"IEND", b""))


PNG_CASES = [(0, d) for d in (1, 2, 4, 8, 16)] + [(2, 8), (2, 16)] + [(3, d) for d in (1, 2, 4, 8)] + [(4, 8), (4, 16), (6, 8), (6, 16)]
CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


@needs_ref
def test_png_every_colour_type_depth_interlacing_and_key(tmp_path):
    rng = np.random.default_rng(11)
    os.mkdir(tmp_path / "shapes"), os.mkdir(tmp_path / "textures")
    write_ply(tmp_path / "shapes/tri.ply", rng.uniform(-1, 1, (3, 3)), triangles=[[0, 1, 2]])
    textures = []
    for color, depth in PNG_CASES:
        for interlace in (False, True):
            for keyed in ((False, True) if color in (0, 2, 3) else (False,)):
                for w, h in ((13, 7), (1, 1), (8, 9), (3, 2)):
                    ch = CHANNELS[color]
                    samples = rng.integers(0, 1 << depth, (h, w, ch))
                    palette = trns = None
                    if color == 3:
                        n = int(rng.integers(max(1, (1 << depth) // 2), (1 << depth) + 1))
                        palette = rng.integers(0, 256, (n, 3))
                        samples = rng.integers(0, n, (h, w, 1))
                        if keyed:
                            trns = bytes(rng.integers(0, 256, int(rng.integers(1, n + 1))).astype("u1"))
                    elif keyed:  # the key = one of the image's own samples, with a noisy high byte for depth < 16
                        pick = samples[int(rng.integers(0, h)), int(rng.integers(0, w))]
                        samples[rng.integers(0, h, 3), rng.integers(0, w, 3)] = pick
                        trns = b"".join(struct.pack(">H", int(v) | (0x3300 if depth == 8 else 0)) for v in pick)
                    name = f"t{len(textures)}_{color}_{depth}_{int(interlace)}_{int(keyed)}_{w}x{h}.png"
                    extra = ((b"gAMA", struct.pack(">I", 45455)), (b"tEXt", b"k\0v")) if w == 13 else ()
                    write_png(tmp_path / "textures" / name, samples, color, depth, interlace, palette, trns, rng, extra)
                    textures.append({"name": name, "uri": "textures/" + name, "nearest": bool(w & 1), "clamp": bool(h & 1)})
    doc = {"asset": {"version": "5.0"}, "shapes": [{"uri": "shapes/tri.ply"}], "textures": textures}
    got, _, _ = both(write_scene(tmp_path, doc))
    assert len(got.textures) == len(textures) >= 150 and not got.textures["is_float"].any() and not got.textures["linear"].any()
    assert got.pixelsb[:, 3].min() == 0  # keys did hit


def test_png_errors(tmp_path):
    os.mkdir(tmp_path / "shapes"), os.mkdir(tmp_path / "textures")
    write_ply(tmp_path / "shapes/tri.ply", np.eye(3), triangles=[[0, 1, 2]])
    doc = {"asset": {"version": "4.2"}, "shapes": [{"uri": "shapes/tri.ply"}], "textures": [{"uri": "textures/t.png"}]}
    path = write_scene(tmp_path, doc)
    rng = np.random.default_rng(3)
    good = tmp_path / "textures/good.png"
    write_png(good, rng.integers(0, 256, (4, 4, 3)), 2, 8)
    data = open(good, "rb").read()
    at = data.index(b"IDAT") + 6  # inside the first IDAT's deflate stream
    for bad, message in ((data[:40], "truncated|bad chunk"), (b"JUNK" + data[4:], "not a PNG|unknown image type"), (data[:8] + data[33:], "IHDR"),
                         (data.replace(b"IDAT", b"IDAX"), "not known|no IDAT"), (data[:at] + b"\xff" * 8 + data[at + 8:], "zlib|pixels|filter")):
        open(tmp_path / "textures/t.png", "wb").write(bad)
        with pytest.raises(yt.YthipError, match=message):
            yt.load_scene_file(path)


# ---------------------------------------------------------------------------------------------------
# Radiance HDR
# ---------------------------------------------------------------------------------------------------
def write_hdr(path, rgbe, layout, header=b"#?RADIANCE\n# made by a test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n"):
    h, w, _ = rgbe.shape
    body = b""
    if layout == "flat":
        body = rgbe.astype("u1").tobytes()
    else:
        rng = np.random.default_rng(w * 31 + h)
        for y in range(h):
            body += bytes([2, 2, w >> 8, w & 255])
            for k in range(4):
                col, i = rgbe[y, :, k].astype("u1"), 0
                while i < w:
                    run = 1
                    while i + run < w and run < 127 and col[i + run] == col[i]:
                        run += 1
                    if run >= 3 or (run > 1 and rng.integers(0, 2)):
                        body += bytes([128 + run, col[i]])
                        i += run
                    else:
                        n = min(int(rng.integers(1, 12)), w - i)
                        body += bytes([n]) + col[i:i + n].tobytes()
                        i += n
    open(path, "wb").write(header + f"-Y {h} +X {w}\n".encode() + body)


@needs_ref
def test_hdr_layouts(tmp_path):
    rng = np.random.default_rng(21)
    os.mkdir(tmp_path / "shapes"), os.mkdir(tmp_path / "textures")
    write_ply(tmp_path / "shapes/tri.ply", rng.uniform(-1, 1, (3, 3)), triangles=[[0, 1, 2]])
    textures = []

    def add(name, rgbe, layout, **kw):
        write_hdr(tmp_path / "textures" / name, rgbe, layout, **kw)
        textures.append({"name": name, "uri": "textures/" + name, "linear": False})

    def image(w, h):
        rgbe = rng.integers(0, 256, (h, w, 4))
        rgbe[..., 3] = rng.integers(100, 150, (h, w))
        rgbe[rng.integers(0, h, 4), rng.integers(0, w, 4), 3] = 0  # exponent 0: black
        rgbe[:, : w // 2, 1] = 77  # long runs
        return rgbe

    add("rle.hdr", image(64, 9), "rle")
    add("rle_wide.hdr", image(300, 3), "rle")
    add("narrow_flat.hdr", image(7, 5), "flat")  # width < 8: never run-length
    wide_flat = image(40, 6)
    wide_flat[0, 0, :3] = (200, 10, 10)  # first bytes are not 2, 2: stb falls back to flat reading
    add("wide_flat.hdr", wide_flat, "flat")
    add("rgbe_signature.hdr", image(16, 2), "rle", header=b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n")
    doc = {"asset": {"version": "4.2"}, "shapes": [{"uri": "shapes/tri.ply"}], "textures": textures,
           "environments": [{"emission": [1, 1, 1], "emission_tex": 0}]}
    got, _, _ = both(write_scene(tmp_path, doc))
    assert got.textures["is_float"].all() and got.textures["linear"].all()  # load_texture overrides the json's "linear"
    assert np.all(got.pixelsf[:, 3] == 1)


def test_hdr_errors(tmp_path):
    os.mkdir(tmp_path / "shapes"), os.mkdir(tmp_path / "textures")
    write_ply(tmp_path / "shapes/tri.ply", np.eye(3), triangles=[[0, 1, 2]])
    path = write_scene(tmp_path, {"asset": {"version": "4.2"}, "shapes": [{"uri": "shapes/tri.ply"}], "textures": [{"uri": "textures/t.hdr"}]})
    rgbe = np.full((2, 16, 4), 128)
    for header, message in ((b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n", "unsupported HDR format"), (b"P6\n", "not a Radiance HDR"),):
        write_hdr(tmp_path / "textures/t.hdr", rgbe, "rle", header=header)
        with pytest.raises(yt.YthipError, match=message):
            yt.load_scene_file(path)
    write_hdr(tmp_path / "textures/t.hdr", rgbe, "rle")
    data = open(tmp_path / "textures/t.hdr", "rb").read()
    open(tmp_path / "textures/t.hdr", "wb").write(data.replace(b"+X 16", b"+X 17"))
    with pytest.raises(yt.YthipError, match="scanline length"):
        yt.load_scene_file(path)
    open(tmp_path / "textures/t.hdr", "wb").write(data.replace(b"-Y 2 +X 16", b"+X 16 -Y 2"))
    with pytest.raises(yt.YthipError, match="data layout"):
        yt.load_scene_file(path)
