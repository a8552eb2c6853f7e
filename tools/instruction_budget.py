#!/usr/bin/env python
"""An instruction budget for the headline workload: wave-instructions per 64 samples of the SHIPPED
kernels, split by what they are spent on — measured, by running kernels of increasing completeness on the same frame under
`rocprofv3 --pmc` (counters only) and differencing:

    k_camera_rays              sample_camera + the pixel's 4 PCG draws                            (per camera ray)
    k_intersect_batch (wide)   the wide walk of those very rays: node steps + leaf tests          (per camera ray; / steps)
    k_trace<falsecolor>        + trace_sample's frame: regenerate, shading point (position, normal), accumulate
    k_trace<eyelight>          + eval_material, 4 bounces of straight-through rays on a hit
    k_trace<naive>             + emission / environment, bsdf sampling, russian roulette (no light sampling)
    k_trace<path>              the default sampler (bench.py's kernel): + light sampling, mis weights
    k_trace<path>, fastmath=1  the same loop with the fast arithmetic: what IEEE division / glibc libm cost
    k_trace<path>, fastmath=2  ... and the own tree's walk instead of the reference tree's

    python tools/instruction_budget.py [SCENE] [RES] [SPP]         (default: plane 1280 64 = BASELINE configs[1])

Prints one row per kernel: VALU / SALU / vector-memory / LDS / scalar-memory / branch wave-instructions per 64 samples (one
wavefront's worth), active lanes per VALU instruction, and the differences between consecutive rows."""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd"))

PASSES = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVES", "GRBM_GUI_ACTIVE"],
          ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_INST_ANY"]]
SAMPLERS = ["falsecolor", "eyelight", "naive", "path"]


def worker(scene, res, spp):
    """The launches, in a fixed order (the parent reads the counters by dispatch order and kernel name)."""
    import numpy as np  # noqa: F401
    import ythip as yt
    os.environ["SCENE"] = scene
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import scenes as ysc
    flat = {"plane": ysc.plane_scene, "cfg4": ysc.instanced_scene, "cfg5": ysc.hair_scene}.get(scene)
    if flat is None:
        import parity as P
        flat = {"cornell1m": P.scene_cornell_1m}.get(scene) or P.SCENES[scene]
    flat = flat()
    ctx = yt.Context(0)
    ctx.upload_scene(flat); ctx.make_trace_bvh(flat); ctx.make_trace_lights(flat)
    ctx.set_pixel_pool(0)  # (plain launches: one workgroup per tile, as the headline runs)
    p = yt.trace_params(sampler="path", resolution=res, samples=1 << 30, batch=spp)
    ctx.make_trace_state(flat, p)
    rays = ctx.camera_rays(p)          # k_camera_rays: one ray per pixel (sample 0)
    ctx.set_profiling(2); ctx.reset_stats()
    ctx.intersect_batch(rays)          # the counting launch (binary walk): the reference's work counts for these rays
    cnt = ctx.get_stats(); ctx.set_profiling(0)
    ctx.intersect_batch(rays)          # k_intersect_batch<false, true>: the wide walk
    ctx.make_own_bvh(flat)
    ctx.intersect_batch_own(rays)      # yt_own::k_intersect_batch<false, true>: the own tree's walk
    print(f"BUDGET rays {len(rays)} nodes {cnt['nodes']} triangles {cnt['triangles']} instances {cnt['instances']}", flush=True)
    for fast in (0, 1, 2):
        for s in SAMPLERS if not fast else ["path"]:
            q = yt.trace_params(sampler=s, resolution=res, samples=1 << 30, batch=spp, fastmath=fast)
            ctx.make_trace_state(flat, q)
            ctx.trace_samples(q)       # warm-up: the 1 + (spp - 1) order probe, tile costs
            ctx.trace_samples(q)       # the measured launch (the LAST k_trace dispatch of this sampler)
    ctx.close()


def collect(scene, res, spp):
    prof = "/opt/rocm/bin/rocprofv3"
    per = {}  # kernel -> counter -> [values in dispatch order]
    line = ""
    for counters in PASSES:
        out = tempfile.mkdtemp(prefix="ythip_budget_", dir="/tmp")
        cmd = [prof, "--pmc"] + counters + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                                            "--worker", scene, str(res), str(spp)]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=300)
        line = next((l for l in r.stdout.splitlines() if l.startswith("BUDGET ")), line)
        rows = {}
        for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                d = rows.setdefault((int(row["Dispatch_Id"]), k), {})
                d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        shutil.rmtree(out, ignore_errors=True)
        if not rows:
            sys.exit(f"no counters (rc {r.returncode}): {r.stderr[-300:]}")
        for (did, k), d in sorted(rows.items()):
            for c, v in d.items():
                per.setdefault(k, {}).setdefault(c, []).append(v)
    return per, line


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    scene = sys.argv[1] if len(sys.argv) > 1 else "plane"
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    per, line = collect(scene, res, spp)
    info = dict(zip(line.split()[1::2], (int(x) for x in line.split()[2::2]))) if line else {}
    nrays = info.get("rays", 0)
    print(f"# instruction budget: {scene} {res} x {spp} spp; {nrays} camera rays per sample; per camera ray the reference's walk pops "
          f"{info.get('nodes', 0) / max(nrays, 1):.2f} nodes, tests {info.get('triangles', 0) / max(nrays, 1):.2f} triangles")
    print("# wave-instructions per 64 samples (= per wavefront-sample); lanes = active lanes per VALU instruction\n")
    hdr = f"{'kernel':58s} {'VALU':>8s} {'SALU':>8s} {'VMEM rd':>8s} {'VMEM wr':>8s} {'LDS':>8s} {'SMEM':>8s} {'branch':>8s} {'lanes':>6s}"
    print(hdr)
    rows = []

    def row(label, k, which, units):
        c = per.get(k)
        if not c:
            print(f"{label:58s} (kernel {k} not seen)")
            return None
        g = lambda n: c[n][which] if n in c and len(c[n]) > (which if which >= 0 else 0) else 0.0  # noqa: E731
        f = 64.0 / units
        v = dict(VALU=g("SQ_INSTS_VALU") * f, SALU=g("SQ_INSTS_SALU") * f, RD=g("SQ_INSTS_VMEM_RD") * f, WR=g("SQ_INSTS_VMEM_WR") * f,
                 LDS=g("SQ_INSTS_LDS") * f, SMEM=g("SQ_INSTS_SMEM") * f, BR=g("SQ_INSTS_BRANCH") * f,
                 lanes=g("SQ_THREAD_CYCLES_VALU") / max(g("SQ_ACTIVE_INST_VALU"), 1))
        print(f"{label:58s} {v['VALU']:8.1f} {v['SALU']:8.1f} {v['RD']:8.1f} {v['WR']:8.1f} {v['LDS']:8.1f} {v['SMEM']:8.1f} {v['BR']:8.1f} {v['lanes']:6.1f}")
        rows.append((label, v))
        return v

    npix = nrays
    row("k_camera_rays (sample_camera + 4 PCG draws), per ray", next((k for k in per if "k_camera_rays" in k), "?"), -1, npix)
    ib = next((k for k in per if k.startswith("yt::k_intersect_batch<false, true>")), "?")
    row("k_intersect_batch, wide walk of the camera rays, per ray", ib, -1, npix)
    ibo = next((k for k in per if k.startswith("yt_own::k_intersect_batch")), "?")
    row("k_intersect_batch, OWN walk of the camera rays, per ray", ibo, -1, npix)
    ktr = [k for k in per if "k_trace" in k]
    # the k_trace kernels in launch order: falsecolor, eyelight, naive, path, path(fast); each launched >= 2 times, the last is measured
    order = sorted(ktr, key=lambda k: 0)  # (dict order = first dispatch order)
    names = SAMPLERS + ["path, fastmath = 1", "path, fastmath = 2 (own tree)"]
    for label, k in zip(names, order):
        row(f"k_trace {label}: {k.replace('yt::', '').replace('yt_fast::', 'fast::').replace('yt_own::', 'own::')}", k, -1, npix * spp)
    print("\n# differences between consecutive k_trace rows (what each stage adds per 64 samples)")
    tr = [r for r in rows if r[0].startswith("k_trace")]
    for (la, a), (lb, b) in zip(tr, tr[1:]):
        print(f"{lb.split(':')[0]:40s} - {la.split(':')[0]:28s} VALU {b['VALU'] - a['VALU']:+9.1f}  SALU {b['SALU'] - a['SALU']:+8.1f}  VMEM rd {b['RD'] - a['RD']:+7.1f}")


if __name__ == "__main__":
    main()
