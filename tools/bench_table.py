#!/usr/bin/env python
"""The workload table of DESIGN.md §5 from ONE bench.py line (the tables and the driver line used to be
different runs):   python tools/bench_table.py profiles/r04_bench.json  [previous.json]
Prints a markdown table: Msamples/s (previous round in parentheses), ms / step, the four fractions, lane utilisation,
wait share, L2 hit rate, TA cycles per wave-level load; bit-exact rows first, then the tolerance-mode rows, the own-tree rows, the streamed rows and the NEE samplers."""
import json
import sys

j = json.load(open(sys.argv[1]))
prev = {}
if len(sys.argv) > 2:
    p = json.load(open(sys.argv[2]))
    prev["configs1"] = p["value"]
    for k, e in enumerate(p.get("other_configs", [])):  # (rounds 1-3 had no names: cfg2b, configs[3], configs[4], cornell9m in this order)
        if "value" in e and e.get("mode", "bit-exact").startswith("bit-exact"):
            prev[e.get("name") or (["cfg2b", "configs3", "configs4", "cornell9m"] + [None] * 16)[k]] = e["value"]
NAMES = {"configs1": "configs[1] 1M-triangle plane, 1280×720×64", "cfg2b": "cfg2b 1M-triangle Cornell box, 1024²×64",
         "configs3": "configs[3] 10 k instances, 1920×1080×256", "configs4": "configs[4] make_hair, 800 k segments, 1280×720×64",
         "cornell9m": "9M-triangle Cornell box (> Infinity Cache), 1024²×16", "materials1": "corpus materials1 (class 3), 1280×533×64",
         "features1": "corpus features1 (general class), 1280×533×64"}


def row(name, value, ms, r, mode, speed=None):
    f = r.get("fractions") or {}
    was = f" ({prev[name]:,.0f})" if mode == "exact" and name in prev else ""
    sp = f" ×{speed:.2f}" if speed else ""
    bound = max(f, key=f.get) if f else "-"
    cells = [NAMES.get(name, name) + {"exact": "", "fast": " — tolerance mode", "own": " — own tree", "stream": " — streaming scheduler (bit-exact)",
                                         "pathdirect": " — sampler pathdirect (bit-exact)", "pathmis": " — sampler pathmis (bit-exact)", "own-stream": " — own tree on the streaming scheduler", "direct-stream": " — sampler pathdirect on the streaming scheduler (bit-exact; × over the fused pathdirect)"}[mode], f"**{value:,.0f}**{was}{sp}", f"{ms:.2f}"]
    cells += [("**%.2f**" % f[k]) if k == bound else ("%.2f" % f[k]) if k in f else "-" for k in ("hbm", "l2", "valu", "ta")]
    cells += ["%.2f" % r["lane_utilisation"] if "lane_utilisation" in r else "-", "%.2f" % r["wave_wait_share"] if "wave_wait_share" in r else "-",
              "%d %%" % round(100 * r["l2_hit_rate"]) if "l2_hit_rate" in r else "-",
              "%.1f" % r["ta_cycles_per_wave_load"] if "ta_cycles_per_wave_load" in r else "-"]
    return "| " + " | ".join(cells) + " |"


print("| workload | Msamples/s (previous round) | ms / step | hbm | l2 | valu | ta | lanes | waiting | L2 hit | TA cycles / wave load |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
print(row("configs1", j["value"], j["ms_per_step"], j.get("roofline", {}), "exact"))
for mode in ("bit-exact", "tolerance", "own-tree", "stream", "pathdirect", "pathmis", "own-stream", "direct-stream"):
    for e in j.get("other_configs", []):
        if "value" in e and e["mode"].startswith(mode):
            print(row(e["name"], e["value"], e["ms_per_step"], e["roofline"], {"bit-exact": "exact", "tolerance": "fast", "own-tree": "own"}.get(mode, mode),
                      e.get("speedup_over_bit_exact")))
c = j.get("cpu_baseline") or {}
if "value" in c:
    h = c.get("host") or {}
    print(f"\nCPU reference (oracle/_ref) on configs[1]: {c['value']:.1f} Msamples/s with {c['cores']} CPUs of the lease "
          f"({h.get('cpu_model', '?')}, {h.get('nproc', '?')} logical CPUs, cgroup cpu.max `{h.get('cgroup_cpu_max', '?')}`); "
          f"affinity sweep: " + ", ".join(f"{e['cpus']} → {e['Msamples_per_s']:.1f}" for e in c.get("sweep", [])) + ".")
