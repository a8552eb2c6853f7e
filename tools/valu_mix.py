#!/usr/bin/env python
"""The VALU instruction mix of the dominant kernels, weighted by loop depth, and the
cycles per wave64 VALU instruction it implies with the per-class costs measured by
tools/microbench/valu_calib.hip at 4 waves per SIMD (gpurun_out/valu_calib.txt).
Writes profiles/r02_valu_calib.json (read by bench.py for the `valu` roofline fraction).

    python tools/valu_mix.py [gpurun_out/valu_calib.txt]

Static estimate: an instruction inside the main loop of the persistent kernel counts 1,
inside the traversal loops (depth >= 2) 8 — the walk runs 10-50 node steps per bounce."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
calib_txt = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "valu_calib.txt")

# measured cycles per wave-instruction per SIMD at 4 waves/SIMD
measured = {}
for line in open(calib_txt):
    m = re.match(r"(.+?)\s+waves/SIMD (\d)\s+.*-> ([0-9.]+) cycles per wave-instruction", line)
    if m and m.group(2) == "4":
        measured[m.group(1).strip()] = float(m.group(3))
cost = {
    "f32 add/sub/mul/fma": measured["v_mul_f32 / v_add_f32"],
    "compare / select / min / max / integer / move": measured["v_cmp_lt_f32 / v_cndmask_b32"],
    "transcendental (rcp, sqrt, exp, log, sin, cos)": measured["v_rcp_f32"],
    "packed f32": measured["v_pk_mul_f32 (2 flop/lane)"],
    "f64": 2 * measured["v_mul_f32 / v_add_f32"],  # (not measured: the f64 vector rate is half the f32 rate)
}


def classify(op):
    if op.startswith("v_pk_"):
        return "packed f32"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op):
        return "transcendental (rcp, sqrt, exp, log, sin, cos)"
    if op.endswith("_f64") or "_f64_" in op:
        return "f64"
    if re.match(r"v_(add|sub|subrev|mul|fma|fmac|mac|mad|fmaak|fmamk)_f32", op) or op in ("v_div_fmas_f32", "v_div_fixup_f32", "v_div_scale_f32"):
        return "f32 add/sub/mul/fma"
    return "compare / select / min / max / integer / move"


asm = "/tmp/ythip_mix.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                "-DYT_DEV_ONLY_PATH", "-S", "--cuda-device-only", "-o", asm,
                os.path.join(ROOT, "yocto-gl_amd", "csrc", "ythip.hip")], check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
kernels = {"k_trace<path, no area lights, wide, matte> (configs[1], configs[3])": "_ZN2yt7k_traceILi0ELi0ELb0ELb1ELb1E",
           "k_trace<path, area lights, wide, matte> (cfg2b)": "_ZN2yt7k_traceILi0ELi2ELb0ELb1ELb1E",
           "k_trace<path, no area lights, wide, general> (configs[4])": "_ZN2yt7k_traceILi0ELi0ELb0ELb1ELb0E"}
out = {"source": "tools/microbench/valu_calib.hip on MI355X (gpurun_out/valu_calib.txt), 4 waves per SIMD",
       "cycles_per_instruction_by_class": cost, "kernels": {}}
tot = collections.Counter()
for label, sym in kernels.items():
    st = [i for i, l in enumerate(lines) if l.startswith(sym)][0]
    en = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end") and i > st][0]
    depth, mix = 0, collections.Counter()
    for l in lines[st:en]:
        if l.startswith(".LBB"):
            m = re.search(r"Depth=(\d+)", l)
            depth = int(m.group(1)) if m else 0
            continue
        t = l.strip().split(" ")[0] if l.strip() and not l.strip().startswith(";") else ""
        if t.startswith("v_") and depth >= 1 and not t.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            mix[classify(t)] += 1 if depth == 1 else 8
    n = sum(mix.values())
    cpi = sum(mix[c] * cost[c] for c in mix) / n
    out["kernels"][label] = {"mix": {c: round(v / n, 4) for c, v in mix.items()}, "cycles_per_instruction": round(cpi, 3)}
    tot.update(mix)
n = sum(tot.values())
out["mix_cycles_per_instruction"] = round(sum(tot[c] * cost[c] for c in tot) / n, 3)
out["peak_wave_instructions_per_simd_cycle"] = round(1 / out["mix_cycles_per_instruction"], 4)
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_valu_calib.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
