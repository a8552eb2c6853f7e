"""The file formats either side of the path, `-m "not gpu"`: the reference's
trace_params JSON files and PLY shapes through libythip's own readers / writers (csrc/yt_io.hip),
checked against the compiled reference where it is present (oracle/_ref: save_trace_params /
update_trace_params / load_scene), and against hand-written files for what the reference's writer
never produces (ascii, big endian, n-gons, lines, points, uchar colours)."""
import json
import os
import struct

import numpy as np
import pytest

import parity as P
from parity import ry, yt

needs_ref = pytest.mark.skipif(not P.have_ref(), reason="oracle/_ref not present")
FIELDS = ["camera", "resolution", "sampler", "falsecolor", "samples", "bounces", "clamp", "nocaustics", "envhidden",
          "tentfilter", "seed", "embreebvh", "highqualitybvh", "noparallel", "pratio", "denoise", "batch"]


def _fields(p):
    return {k: getattr(p, k) for k in FIELDS}


def test_trace_params_json_round_trip_and_update_semantics():
    d = yt.params_from_json("{}")
    assert _fields(d) == _fields(yt.trace_params())  # trace_params{} (yocto_trace.h:95-113)
    p = yt.trace_params(sampler="pathmis", falsecolor="element", resolution=720, samples=33, bounces=5, clamp=2.5,
                        nocaustics=1, envhidden=1, tentfilter=1, seed=(1 << 63) + 12345, highqualitybvh=1, pratio=4,
                        denoise=1, batch=7, camera=2)
    text = yt.params_to_json(p)
    obj = json.loads(text)
    assert list(obj) == FIELDS  # the reference's keys, in its order
    assert obj["sampler"] == "pathmis" and obj["falsecolor"] == "element" and obj["seed"] == (1 << 63) + 12345
    assert obj["nocaustics"] is True and obj["embreebvh"] is False and obj["clamp"] == 2.5
    assert _fields(yt.params_from_json(text)) == _fields(p)
    # update: absent keys keep what is there; unknown keys and nested values are ignored
    q = yt.params_from_json('{"samples": 9, "whatever": {"a": [1, 2, {"b": "}"}]}, "sampler": "naive"}', yt.trace_params(bounces=3))
    assert q.samples == 9 and q.sampler == yt.SAMPLERS.index("naive") and q.bounces == 3
    for bad in ['{"sampler": "nope"}', '{"samples": "many"}', '{"nocaustics": 1}', '[1]', '{"samples" 3}']:
        with pytest.raises(yt.YthipError):
            yt.params_from_json(bad)
    # this library's one extra field, the tolerance-mode switch: off = a file the reference's own keys describe completely;
    # on = one more key, which the reference's from_json ignores (json.value(key, default) per known key)
    assert d.fastmath == 0 and "fastmath" not in obj
    f = yt.trace_params(fastmath=1, samples=3)
    fobj = json.loads(yt.params_to_json(f))
    assert list(fobj) == FIELDS + ["fastmath"] and fobj["fastmath"] is True
    assert yt.params_from_json(yt.params_to_json(f)).fastmath == 1
    assert yt.params_from_json('{"fastmath": false}', yt.trace_params(fastmath=1)).fastmath == 0
    with pytest.raises(yt.YthipError):
        yt.params_from_json('{"fastmath": 1}')


@needs_ref
def test_trace_params_files_interoperate_with_the_reference(tmp_path):
    """Files written by the reference's save_trace_params are read identically; files written here are
    read identically by the reference's update_trace_params — all samplers and falsecolor modes."""
    import ctypes as C
    lib = ry.lib()
    for k in range(18):
        p = yt.trace_params(sampler=k % 9, falsecolor=k, resolution=100 + k, samples=k + 1, bounces=k % 5, clamp=0.1 * k + 0.3,
                            nocaustics=k & 1, envhidden=(k >> 1) & 1, tentfilter=(k >> 2) & 1, seed=961748941 + k * 7919,
                            embreebvh=k % 3 == 0, highqualitybvh=k % 2, noparallel=k % 4 == 1, pratio=1 + k % 8,
                            denoise=k % 5 == 2, batch=1 + k, camera=k % 3)
        theirs = str(tmp_path / f"ref_{k}.json")
        assert lib.ref_params_save(C.byref(p), theirs.encode()) == 0, lib.ref_load_error()
        assert _fields(yt.params_from_json(open(theirs).read())) == _fields(p)
        ours = str(tmp_path / f"ours_{k}.json")
        open(ours, "w").write(yt.params_to_json(p))
        back = yt.trace_params()
        assert lib.ref_params_load(ours.encode(), C.byref(back)) == 0, lib.ref_load_error()
        assert _fields(back) == _fields(p)


def _same_shape(mine, flat, what):
    for name in ["positions", "normals", "texcoords", "colors", "radius", "points", "lines", "triangles", "quads"]:
        a, b = mine[name], getattr(flat, name)
        assert a.shape[0] == b.shape[0], f"{what}: {name} count {a.shape[0]} != {b.shape[0]}"
        assert a.tobytes() == np.ascontiguousarray(b).tobytes(), f"{what}: {name}"


@needs_ref
def test_ply_reader_equals_the_reference_loader_on_its_own_test_shapes():
    """Every .ply under the reference's tests/_version43 (binary little endian; triangles and quads, normals,
    texcoords): the arrays of ythip_ply_read are the bytes of the reference's load_scene of that file."""
    root = os.path.join(os.environ.get("YOCTO_REF", "/root/reference"), "tests")
    files = []
    for d, _, fs in os.walk(root):
        files += [os.path.join(d, f) for f in fs if f.endswith(".ply")]
    if not files:
        pytest.skip("the reference's test shapes are not present")
    assert len(files) >= 5
    for f in sorted(files)[:60]:
        flat = ry.RefScene.load(f).flat()
        assert len(flat.shapes) == 1
        _same_shape(yt.load_ply(f, flip_texcoord=True), flat, os.path.basename(f))


ASCII = """ply
format ascii 1.0
comment hand-written: n-gons, a line strip, points, uchar colours, double coordinates
element vertex 6
property double x
property double y
property float z
property float nx
property float ny
property float nz
property float s
property float t
property uchar red
property uchar green
property uchar blue
property float radius
element face 5
property list uchar int vertex_indices
element line 2
property list uchar uint vertex_indices
element point 2
property list uchar short vertex_indices
end_header
0 0 0.125 0 0 1 0 0 255 0 0 0.01
1 0 1e-3 0 0 1 1 0 0 255 0 0.02
1 1 0.3333333333 0 0 1 1 1 0 0 255 0.03
0 1 -2.5 0 0 1 0 1 10 20 30 0.04
0.5 1.5 7 0 1 0 0.5 0.25 1 2 3 0.05
-0.1 0.7 0.1 1 0 0 0.1 0.9 200 100 50 0.06
3 0 1 2
5 0 2 3 4 5
2 4 5
1 3
4 0 1 2 3
4 0 1 2 3
2 4 5
1 0
2 1 2
"""


def _binary_from_ascii(text, fmt):
    """The same file as binary_little_endian / binary_big_endian (independent writer for the test)."""
    lines = text.strip().split("\n")
    k = lines.index("end_header")
    header = [("format " + fmt + " 1.0") if l.startswith("format") else l for l in lines[:k + 1]]
    e = "<" if fmt == "binary_little_endian" else ">"
    code = {"char": "b", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "uint": "I", "float": "f", "double": "d"}
    elems, cur = [], None
    for l in header:
        t = l.split()
        if t[0] == "element":
            cur = [int(t[2]), []]
            elems.append(cur)
        elif t[0] == "property":
            cur[1].append(("list", t[3]) if t[1] == "list" else ("scalar", t[1]))
    body, rows = b"", lines[k + 1:]
    for count, props in elems:
        for _ in range(count):
            vals = rows.pop(0).split()
            for kind, ty in props:
                if kind == "list":
                    n = int(vals.pop(0))
                    body += struct.pack("B", n)
                    for _ in range(n):
                        body += struct.pack(e + code[ty], int(vals.pop(0)))
                else:
                    v = vals.pop(0)
                    body += struct.pack(e + code[ty], float(v) if ty in ("float", "double") else int(v))
    return ("\n".join(header) + "\n").encode() + body


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
def test_ply_reader_on_a_hand_written_file(fmt, tmp_path):
    path = str(tmp_path / "shape.ply")
    open(path, "wb").write(ASCII.encode() if fmt == "ascii" else _binary_from_ascii(ASCII, fmt))
    s = yt.load_ply(path, flip_texcoord=True)
    # a face with four corners makes every face a quad: triangle -> (a, b, c, c), pentagon fanned
    assert s["quads"].tolist() == [[0, 1, 2, 2], [0, 2, 3, 3], [0, 3, 4, 4], [0, 4, 5, 5], [4, 5, -1, -1], [3, -1, -1, -1],
                                   [0, 1, 2, 3]]
    assert len(s["triangles"]) == 0
    assert s["lines"].tolist() == [[0, 1], [1, 2], [2, 3], [4, 5]]
    assert s["points"].reshape(-1).tolist() == [0, 1, 2]
    assert s["positions"][2].tolist() == [1.0, 1.0, float(np.float32(0.3333333333))]
    assert s["positions"][1][2] == np.float32(1e-3) and s["positions"][3][2] == np.float32(-2.5)
    assert s["texcoords"][4].tolist() == [0.5, 0.75]  # v flipped
    assert s["colors"][0].tolist() == [255.0, 0.0, 0.0, 1.0]  # the reference casts uchar colours, alpha 1
    assert np.allclose(s["radius"].reshape(-1), [0.01, 0.02, 0.03, 0.04, 0.05, 0.06])
    if P.have_ref():
        _same_shape(s, ry.RefScene.load(path).flat(), fmt)


def test_ply_faces_without_a_quad_become_triangles(tmp_path):
    text = ASCII.replace("element face 5", "element face 4").replace("4 0 1 2 3\n4 0 1 2 3\n", "4 0 1 2 3\n", 1)
    path = str(tmp_path / "tris.ply")
    open(path, "w").write(text)
    s = yt.load_ply(path)
    assert s["triangles"].tolist() == [[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 4, 5], [4, 5, -1], [3, -1, -1]]
    assert len(s["quads"]) == 0
    if P.have_ref():
        _same_shape(s, ry.RefScene.load(path).flat(), "triangles")


def test_ply_reader_errors(tmp_path):
    bad = str(tmp_path / "bad.ply")
    open(bad, "w").write("plx\nformat ascii 1.0\nend_header\n")
    with pytest.raises(yt.YthipError):
        yt.load_ply(bad)
    empty = str(tmp_path / "empty.ply")
    open(empty, "w").write("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\nend_header\n0 0 0\n")
    with pytest.raises(yt.YthipError) as e:
        yt.load_ply(empty)
    assert "empty shape" in str(e.value)
    with pytest.raises(yt.YthipError):
        yt.load_ply(str(tmp_path / "missing.ply"))
