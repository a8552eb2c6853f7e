#!/usr/bin/env python
"""make_trace_bvh on many shapes / many instances: wall time of the library (device
builds + host thread pool, device instance tree) against its own one-thread host build and against the
reference's make_trace_bvh on all host cores (oracle/_ref: TEST INFRASTRUCTURE), trees compared byte for byte.
  python tools/build_times.py [instances|shapes]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ythip as yt, scenes as ysc
import refyocto as ry
f32 = np.float32
which = sys.argv[1] if len(sys.argv) > 1 else "instances"
if which == "instances":  # a 1000 x 1000 grid of one 64-triangle sphere: 1,000,000 instances
    flat = ysc.instanced_scene(grid=1000, sphere_steps=(8, 4), radius=0.04)
    label = f"{len(flat.instances):,} instances of one {int(flat.shapes[0]['num_triangles'])}-triangle shape"
else:  # 2,000 shapes of 2,000 triangles each: every one below the device threshold
    rng = np.random.default_rng(3)
    flat = yt.FlatScene()
    flat.add_camera(ysc.lookat_frame((0, 3, 8), (0, 0, 0)))
    m = flat.add_material("matte", color=(0.6, 0.6, 0.6))
    for k in range(2000):
        n = 2000
        p = (rng.random((3 * n, 3)) * 0.2 + rng.integers(-20, 20, 3)).astype(f32)
        flat.add_instance(flat.add_shape(p, triangles=np.arange(3 * n, dtype=np.int32).reshape(n, 3)), m)
    flat.add_environment((1, 1, 1))
    label = f"{len(flat.shapes):,} shapes x 2,000 triangles"
def timed_build(builder, threads):
    if threads is None: os.environ.pop("YTHIP_BUILD_THREADS", None)
    else: os.environ["YTHIP_BUILD_THREADS"] = str(threads)
    ctx = yt.Context(0); ctx.upload_scene(flat); ctx.set_bvh_builder(builder)
    ctx.make_trace_bvh(flat)  # warm-up (code objects, allocations)
    t = time.perf_counter(); ctx.make_trace_bvh(flat); dt = time.perf_counter() - t
    info = ctx.bvh_build_info(); b = ctx.download_bvh(); ctx.close()
    return dt, info, b
t_lib, i_lib, b_lib = timed_build("device", None)
t_one, i_one, b_one = timed_build("host", 0)
print(f"{label}")
print(f"  libythip (device + pool)  {1e3 * t_lib:9.1f} ms   build {i_lib['build_ms']:.1f} + bake {i_lib['bake_ms']:.1f} ms; device trees {i_lib['device_trees']}, "
      f"device instance tree {i_lib['device_tlas']}, host trees {i_lib['host_trees']} on {i_lib['host_threads']} threads")
print(f"  libythip (one host thread){1e3 * t_one:9.1f} ms   build {i_one['build_ms']:.1f} + bake {i_one['bake_ms']:.1f} ms")
print(f"  identical trees: {b_lib.same_as(b_one)}")
if ry.available():
    scene = ry.RefScene.from_flat(flat)
    ry.RefBvh(scene)
    t = time.perf_counter(); rb = ry.RefBvh(scene); t_ref = time.perf_counter() - t
    r = rb.flat()
    print(f"  reference make_trace_bvh  {1e3 * t_ref:9.1f} ms   ({ry.hardware_concurrency()} host threads); identical to the library's: "
          f"{r.nodes.tobytes() == b_lib.nodes.tobytes() and r.primitives.tobytes() == b_lib.primitives.tobytes()}")
