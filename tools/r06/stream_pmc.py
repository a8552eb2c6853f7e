#!/usr/bin/env python
"""Round 6: hardware counters of a streamed batch's kernels next to the fused kernel's ("lanes, ta, valu, spills
and steps over the longest lane reported beside the megakernel's").

  python tools/r06/stream_pmc.py SCENE [SPP]          the parent: two rocprofv3 --pmc passes (bench.py's counter sets; counters only,
                                                      never with tracing) of each of the two children below, summed per kernel
  python tools/r06/stream_pmc.py --child SCENE SPP MODE   a child: MODE fused | stream — one warm-up batch + one batch

Counters are summed over ALL dispatches of a kernel in the child (warm-up included: the ratios do not care), GRBM_GUI_ACTIVE too
(the dispatch windows do not overlap: one group, and the profiler serialises launches).  Printed per kernel: share of the
process' kernel cycles, VALU issue fraction (2 cycles per wave64 instruction on 1024 SIMDs), TA busy fraction, lanes per VALU
instruction, the waiting share of a wavefront's life, L2 hit rate, HBM bytes."""
import csv, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd")); sys.path.insert(0, ROOT)

PASSES = [["FETCH_SIZE", "TCC_REQ_sum", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
          ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_THREAD_CYCLES_VALU",
           "TA_TA_BUSY_sum", "TA_FLAT_READ_WAVEFRONTS_sum", "GRBM_GUI_ACTIVE"]]


def child(scene, spp, mode):
    import ythip as yt, bench
    w = bench._workloads()[scene]
    flat = w["make"]()
    ctx = bench.open_context(0, flat)
    fm = int(os.environ.get("FASTMATH", "0"))  # 1 / 2: the tolerance / own-tree units' kernels (yt_fast:: / yt_own::)
    if fm == 2:
        ctx.make_own_bvh(flat)
    p = yt.trace_params(sampler="path", resolution=w["resolution"], samples=1 << 30, batch=spp, fastmath=fm)
    if mode == "stream":
        ctx.set_scheduler(1)
        o, c = (int(x) for x in os.environ.get("VARIANT", "1:3").split(":"))
        ctx.set_stream_options(order=o, cell_bits=c)
        ctx.set_stream_groups(1)
    ctx.make_trace_state(flat, p)
    ctx.trace_samples(p)
    ctx.trace_samples(p)
    ctx.close()


def collect(scene, spp, mode):
    tot = {}
    for k, counters in enumerate(PASSES):
        out = tempfile.mkdtemp(prefix="ythip_spmc_", dir="/tmp")
        cmd = ["/opt/rocm/bin/rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                                                                 "--child", scene, str(spp), mode]
        env = dict(os.environ, TMPDIR="/tmp", YTHIP_LPT_PROBE="0", YTHIP_BENCH_PROGRESS="0")
        proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            proc.communicate(timeout=int(os.environ.get("PMC_TIMEOUT", "400")))
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, 9)
            proc.communicate()
            shutil.rmtree(out, ignore_errors=True)
            return None
        for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if not name.startswith(("yt::k_trace", "yt::ks_", "yt_own::k_trace", "yt_own::ks_", "yt_fast::k_trace", "yt_fast::ks_")):
                    continue
                c = row["Counter_Name"] + ("#2" if k == 1 and row["Counter_Name"] == "GRBM_GUI_ACTIVE" else "")
                d = tot.setdefault(name, {})
                d[c] = d.get(c, 0.0) + float(row["Counter_Value"])
                if c == "SQ_INSTS_VALU":
                    d["dispatches"] = d.get("dispatches", 0) + 1
        shutil.rmtree(out, ignore_errors=True)
    return tot


def report(tot, title):
    print(f"== {title}")
    cyc_all = sum(d.get("GRBM_GUI_ACTIVE", 0) for d in tot.values()) / 8.0
    for name, d in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        cyc, cyc2 = d.get("GRBM_GUI_ACTIVE", 0) / 8.0, d.get("GRBM_GUI_ACTIVE#2", 0) / 8.0
        if not cyc or not cyc2:
            continue
        valu = d["SQ_INSTS_VALU"] * 2.0 / (1024 * cyc)
        ta = d.get("TA_TA_BUSY_sum", 0) / (256 * cyc2)
        lanes = d.get("SQ_THREAD_CYCLES_VALU", 0) / max(64.0 * d.get("SQ_ACTIVE_INST_VALU", 1), 1)
        wait = d.get("SQ_WAIT_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1)
        hit = d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1)
        hbm = (2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024
        print(f"  {name:58s} share {cyc / cyc_all:5.3f}  dispatches {d.get('dispatches', 0) // 8 or d.get('dispatches', 0):5d}  valu {valu:5.3f}  ta {ta:5.3f}  lanes {lanes:5.3f}  "
              f"useful valu {valu * lanes:5.3f}  waiting {wait:5.3f}  L2 hit {hit:5.3f}  hbm {hbm / 1e9:7.2f} GB  VALU insts {d['SQ_INSTS_VALU'] / 1e9:7.3f} G  wave loads {d.get('TA_FLAT_READ_WAVEFRONTS_sum', 0) / 1e9:6.3f} G")
    print(f"  kernel cycles of the process: {cyc_all / 1e6:.1f} M (shader clock)")


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    scene, spp = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16
    for mode in ("fused", "stream"):
        t = collect(scene, spp, mode)
        if t is None:
            print(f"{scene} {mode}: a rocprofv3 pass timed out")
            continue
        report(t, f"{scene} {spp} spp, {mode}")
