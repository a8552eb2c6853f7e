"""The file readers under AddressSanitizer + UBSan against damaged inputs (`-m "not gpu"`): tests/cpp/io_fuzz.cpp is
built with g++ from the two host-code units of the library (csrc/yt_io.hip, csrc/yt_sceneio.hip) and run on a scene
that exercises every reader path — binary and ascii PLY, triangles / quads / lines / points, PNG in several colour
types incl. palette + tRNS + Adam7 + 16 bit, Radiance HDR run-length and flat, JPEG baseline / progressive / grey / CMYK, OpenEXR
ZIP / RLE / tiled / flat, BMP and TGA — with one file damaged per iteration.
The readers may load or refuse; a sanitizer report, a crash or a hang fails the test."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from test_sceneio import write_hdr, write_ply, write_png

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yocto-gl_amd", "csrc")

ASCII_PLY = """ply
format ascii 1.0
element vertex 5
property double x
property float y
property float z
property float nx
property float ny
property float nz
property float u
property float v
property uchar red
property uchar green
property uchar blue
property uchar alpha
element face 3
property list uchar int vertex_indices
end_header
0 0 0 0 0 1 0 0 255 0 0 255
1 0 0 0 0 1 1 0 0 255 0 255
1 1 0 0 0 1 1 1 0 0 255 128
0 1 0 0 0 1 0 1 9 9 9 0
0.5 2 0 0 0 1 .5 1 1 2 3 4
3 0 1 2
4 0 1 2 3
5 0 1 2 3 4
"""


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("io_fuzz_build")
    exe = str(d / "io_fuzz")
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    cmd = ["g++"] + flags + [os.path.join(ROOT, "tests", "cpp", "io_fuzz.cpp"), "-x", "c++", os.path.join(CSRC, "yt_io.hip"),
                             os.path.join(CSRC, "yt_sceneio.hip"), "-o", exe, "-lz", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode and "sanitize" in r.stderr and "unsupported" in r.stderr:
        pytest.skip("g++ without sanitizer runtimes")
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    d = tmp_path_factory.mktemp("io_fuzz_scene")
    os.mkdir(d / "shapes"), os.mkdir(d / "textures")
    rng = np.random.default_rng(9)
    write_ply(d / "shapes/tri.ply", rng.uniform(-1, 1, (40, 3)), triangles=[[int(a), int(b), int(c)] for a, b, c in rng.integers(0, 40, (30, 3))])
    write_ply(d / "shapes/hair.ply", rng.uniform(-1, 1, (12, 3)), lines=[[0, 1, 2, 3], [4, 5], [6, 7, 8]])
    write_ply(d / "shapes/dots.ply", rng.uniform(-1, 1, (6, 3)), points=[[0, 1], [2], [3, 4, 5]], radius=rng.uniform(0.01, 0.1, 6))
    open(d / "shapes/ascii.ply", "w").write(ASCII_PLY)
    textures = []
    for k, (color, depth, interlace, keyed) in enumerate([(2, 8, False, False), (3, 4, True, True), (0, 16, True, True), (6, 16, False, False),
                                                          (4, 8, True, False), (0, 1, False, False)]):
        ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
        samples, palette, trns = rng.integers(0, 1 << depth, (9, 11, ch)), None, None
        if color == 3:
            palette, samples, trns = rng.integers(0, 256, (12, 3)), rng.integers(0, 12, (9, 11, 1)), bytes(range(5))
        elif keyed:
            trns = int(samples[0, 0, 0]).to_bytes(2, "big")
        write_png(d / "textures" / f"t{k}.png", samples, color, depth, interlace, palette, trns, rng)
        textures.append({"name": f"t{k}", "uri": f"textures/t{k}.png"})
    rgbe = rng.integers(0, 256, (6, 24, 4))
    rgbe[:, :12, 0] = 50
    write_hdr(d / "textures/rle.hdr", rgbe, "rle")
    write_hdr(d / "textures/flat.hdr", rgbe[:, :5], "flat")
    textures += [{"name": "rle", "uri": "textures/rle.hdr"}, {"name": "flat", "uri": "textures/flat.hdr"}]
    # round 5's readers: JPEG (baseline, progressive + restart intervals, grey, CMYK), OpenEXR (ZIP, RLE, tiled, flat), BMP, TGA
    try:
        from PIL import Image as PIL
        import test_imgcodec as tc
        pic = lambda w, h, ch: tc.picture(rng, w, h, ch)
        tc.to_image(pic(40, 24, 3), "RGB").save(d / "textures/base.jpg", "JPEG", quality=70, subsampling=2, restart_marker_blocks=2)
        tc.to_image(pic(33, 17, 3), "RGB").save(d / "textures/prog.jpg", "JPEG", quality=80, progressive=True, subsampling=1)
        tc.to_image(pic(17, 9, 1), "L").save(d / "textures/grey.jpg", "JPEG")
        tc.to_image(pic(16, 16, 4), "CMYK").save(d / "textures/cmyk.jpg", "JPEG", progressive=True)
        tc.to_image(pic(13, 7, 3), "RGB").save(d / "textures/pal.bmp", "BMP")
        tc.to_image(pic(13, 7, 4), "RGBA").save(d / "textures/rgba.bmp", "BMP")
        pal = PIL.frombytes("P", (9, 5), bytes(rng.integers(0, 256, 45, dtype=np.uint8)))
        pal.putpalette([int(v) for v in rng.integers(0, 256, 768)])
        pal.save(d / "textures/p8.bmp", "BMP")
        tc.to_image(pic(12, 10, 4), "RGBA").save(d / "textures/rle.tga", "TGA", rle=True)
        pal.save(d / "textures/cmap.tga", "TGA")
        tc.write_exr(d / "textures/zip.exr", tc.exr_planes(rng, 19, 37, ["A", "B", "G", "R"], "float16"), 3)
        tc.write_exr(d / "textures/rle.exr", tc.exr_planes(rng, 16, 5, ["B", "G", "R"], "float32"), 1, line_order=1)
        tc.write_exr(d / "textures/tiled.exr", tc.exr_planes(rng, 21, 13, ["Y"], "float16"), 2, tile=(8, 8))
        tc.write_exr(d / "textures/flat.exr", tc.exr_planes(rng, 4, 3, ["B", "G", "R"], "uint32"), 0)
        textures += [{"name": n, "uri": "textures/" + n} for n in ("base.jpg", "prog.jpg", "grey.jpg", "cmyk.jpg", "pal.bmp", "rgba.bmp", "p8.bmp",
                                                                    "rle.tga", "cmap.tga", "zip.exr", "rle.exr", "tiled.exr", "flat.exr")]
    except ImportError:
        pass
    doc = {"asset": {"version": "4.2"},
           "cameras": [{"name": "c", "lookat": [0, 0, 3, 0, 0, 0, 0, 1, 0], "lens": 0.05}],
           "textures": textures,
           "materials": [{"name": "m", "type": "glossy", "color": [0.5, 0.5, 0.5], "color_tex": 0, "normal_tex": 3}],
           "shapes": [{"name": n, "uri": f"shapes/{n}.ply"} for n in ("tri", "hair", "dots", "ascii")],
           "instances": [{"name": f"i{k}", "shape": k, "material": 0} for k in range(4)],
           "environments": [{"name": "e", "emission": [1, 1, 1], "emission_tex": 6}]}
    json.dump(doc, open(d / "scene.json", "w"), indent=1)
    # the same files as a format-4.0 scene (named elements, references by name) in a directory of its own
    d40 = tmp_path_factory.mktemp("io_fuzz_scene40")
    shutil.copytree(d / "shapes", d40 / "shapes"), shutil.copytree(d / "textures", d40 / "textures")
    old = {"asset": {"copyright": "fuzz"},
           "cameras": {"c": {"lookat": [0, 0, 3, 0, 0, 0, 0, 1, 0], "ortho": False}},
           "environments": {"e": {"emission": [1, 1, 1], "emission_tex": "rle"}},
           "materials": {"m": {"type": "metallic", "color_tex": "t0", "normal_tex": "t3", "emission_tex": "flat"}},
           "instances": {f"i{k}": {"shape": n, "material": "m"} for k, n in enumerate(("tri", "hair", "dots", "ascii"))}}
    json.dump(old, open(d40 / "scene.json", "w"), indent=1)
    return str(d), str(d40)


def test_the_fuzz_scenes_load_undamaged(scene):
    from parity import yt
    flat, names, _ = yt.load_scene_file(os.path.join(scene[0], "scene.json"))
    assert len(flat.shapes) == 4 and len(flat.textures) in (8, 21) and flat.shapes["num_quads"][3] == 5
    flat, names, _ = yt.load_scene_file(os.path.join(scene[1], "scene.json"))
    assert names["shapes"] == ["tri", "hair", "dots", "ascii"] and names["textures"] == ["rle", "flat", "t0", "t3"]  # (order of first mention: emission_tex before color_tex)


@pytest.mark.parametrize("seed,which", [(1, 0), (2, 0), (3, 1), (4, 1)])
def test_damaged_files_never_crash_the_readers(fuzzer, scene, seed, which):
    scene = scene[which]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    iters = os.environ.get("YT_FUZZ_ITERS", "1200")
    r = subprocess.run([fuzzer, scene, iters, str(seed)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert f"io_fuzz: {iters} iterations" in r.stdout
    loaded = int(r.stdout.split("iterations,")[1].split("loaded")[0])
    assert 0.08 * int(iters) < loaded < 0.92 * int(iters)  # both outcomes are exercised
