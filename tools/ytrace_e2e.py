#!/usr/bin/env python
"""End-to-end check of the drop-in under the reference's own app: BASELINE configs[1]
saved with the reference's save_scene, rendered by oracle/_ref/ytrace_cpu and
oracle/_ref/ytrace_hip (the same UNMODIFIED apps/ytrace.cpp, see oracle/Makefile),
the app's own timer lines printed side by side.  Diagnostic; needs oracle/_ref."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("yocto-gl_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
import refyocto as ry
import scenes as ysc

spp = sys.argv[1] if len(sys.argv) > 1 else "64"
tmp = tempfile.mkdtemp()
fn = os.path.join(tmp, "plane", "plane.json")
os.makedirs(os.path.dirname(fn))
ry.RefScene.from_flat(ysc.plane_scene()).save(fn)
imgs = {}
for exe in ["ytrace_cpu", "ytrace_hip"]:
    out = os.path.join(tmp, exe + ".hdr")
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe), "--scene", fn, "--output", out,
                        "--samples", spp, "--batch", spp, "--resolution", "1280"], capture_output=True, text=True)
    print(f"== {exe} (exit {r.returncode})")
    print("\n".join(l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l))
    imgs[exe] = ry.load_image(out)
a, b = imgs["ytrace_cpu"], imgs["ytrace_hip"]
close = np.isclose(a, b, rtol=1e-3, atol=1e-5).all(-1)
print(f"pixels within 1e-3 relative: {close.mean() * 100:.3f} %   mean {a[..., :3].mean():.6f} vs {b[..., :3].mean():.6f}")
