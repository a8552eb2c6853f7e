#!/usr/bin/env python
"""Recomputes a bench line's roofline from the files beside it and fails if they disagree.

  python tools/check_bench_line.py LINE.json DETAIL.json [KERNEL_STATS.csv] [--tol 0.02]

  LINE.json          the one JSON line `python bench.py` printed
  DETAIL.json        its side file (bench.py --detail): roofline.counters_per_launch = the raw --pmc counters
  KERNEL_STATS.csv   `rocprofv3 --kernel-trace --stats` of the same command in the same session (optional)

Checked, each within `tol` (relative):
  value                  = pixels x spp / ms_per_step
  fractions.valu         = SQ_INSTS_VALU x 2 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
  fractions.hbm, traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch / launch time / 8 TB/s
  fractions.ta           = TA_TA_BUSY_sum / (256 CUs x GRBM_GUI_ACTIVE of ITS pass / 8)
  frac                   = the largest fraction, bound = its name
  lane_utilisation       = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
  the trace's average duration of the line's kernel vs roofline.launch_ms_avg (tracing costs a 5-ms launch up to 3 %:
  this one within max(tol, 0.04))
and printed: the effective shader clock of the profiled launch = GRBM_GUI_ACTIVE / 8 / launch time.
Exit status 0 = consistent."""
import csv
import json
import sys


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


def check(line, detail, stats_rows=None, tol=0.02):
    """Returns (list of failures, dict of recomputed values)."""
    bad, out = [], {}
    r, c = line["roofline"], detail["roofline"]["counters_per_launch"]
    w, h = line["config"]["resolution"]
    units = w * h * line["config"]["spp"]
    out["value"] = units / line["ms_per_step"] / 1e3
    if rel(out["value"], line["value"]) > tol:
        bad.append(f"value {line['value']} vs pixels x spp / ms_per_step {out['value']:.1f}")
    sec = r["launch_ms_avg"] * 1e-3
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    out["shader_clock_GHz"] = cyc / sec / 1e9
    fr = {"valu": c["SQ_INSTS_VALU"] * 2.0 / (1024 * cyc),
          "hbm": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / sec / 8e12,
          "l2": c["TCC_REQ_sum"] * 128 / sec / 34.5e12}
    if "TA_TA_BUSY_sum" in c and "GRBM_GUI_ACTIVE_ta_pass" in c:
        fr["ta"] = c["TA_TA_BUSY_sum"] / (256 * c["GRBM_GUI_ACTIVE_ta_pass"] / 8.0)
    out["fractions"] = fr
    for k, v in fr.items():
        if k in r["fractions"] and rel(v, r["fractions"][k]) > tol:
            bad.append(f"fractions.{k} {r['fractions'][k]} vs counters {v:.4f}")
    b = max(fr, key=fr.get)
    if b != r["bound"] and rel(fr[b], fr.get(r["bound"], 0)) > tol:
        bad.append(f"bound {r['bound']} vs counters {b}")
    if rel(max(fr.values()), r["frac"]) > tol:
        bad.append(f"frac {r['frac']} vs counters {max(fr.values()):.4f}")
    if "traffic" in r and r["traffic"]:
        t = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        if rel(t, r["traffic"]) > tol:
            bad.append(f"traffic {r['traffic']} vs counters {t:.0f}")
    if "lane_utilisation" in r and c.get("SQ_ACTIVE_INST_VALU"):
        lu = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        out["lane_utilisation"] = lu
        if rel(lu, r["lane_utilisation"]) > tol:
            bad.append(f"lane_utilisation {r['lane_utilisation']} vs counters {lu:.4f}")
    if stats_rows is not None:
        want = r["kernel"].replace("void ", "")
        rows = [x for x in stats_rows if x["Name"].split("(")[0].replace("void ", "").strip() == want]
        if not rows:
            bad.append(f"the trace has no row for {want}")
        else:
            avg = float(rows[0]["AverageNs"]) / 1e6
            out["trace_avg_ms"], out["trace_calls"] = avg, int(rows[0]["Calls"])
            if rel(avg, r["launch_ms_avg"]) > max(tol, 0.04):
                bad.append(f"kernel trace average {avg:.4f} ms vs launch_ms_avg {r['launch_ms_avg']}")
    return bad, out


def main(argv):
    tol = 0.02
    if "--tol" in argv:
        k = argv.index("--tol")
        tol = float(argv[k + 1])
        argv = argv[:k] + argv[k + 2:]
    if len(argv) < 3:
        sys.exit(__doc__)
    line = json.loads([l for l in open(argv[1]).read().splitlines() if l.startswith("{")][-1])
    detail = json.load(open(argv[2]))
    rows = list(csv.DictReader(open(argv[3]))) if len(argv) > 3 else None
    bad, out = check(line, detail, rows, tol)
    print(f"value {line['value']} {line['unit']}, {line['ms_per_step']} ms / step; kernel {line['roofline']['kernel']}, launch {line['roofline']['launch_ms_avg']} ms")
    print("recomputed: " + ", ".join(f"{k} {v:.4f}" for k, v in out["fractions"].items()) +
          f"; effective shader clock {out['shader_clock_GHz']:.3f} GHz" +
          (f"; kernel trace average {out['trace_avg_ms']:.4f} ms over {out['trace_calls']} launches" if "trace_avg_ms" in out else ""))
    for b in bad:
        print("MISMATCH: " + b)
    print("consistent" if not bad else f"{len(bad)} mismatches (tolerance {tol})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
