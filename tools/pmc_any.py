#!/usr/bin/env python
"""Any set of hardware counters for a bench workload's k_trace launches:
    python tools/pmc_any.py WORKLOAD[:fast|:own] COUNTER [COUNTER ...] [-- COUNTER ...]      ("--" starts another rocprofv3 pass)
Runs `bench.py --worker WORKLOAD` under rocprofv3 --pmc (counters only), prints the per-launch means of the worker's timed
launches, and a few ratios when their ingredients are there (TA busy share, L1 accesses per VALU instruction ...)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd"))
import bench  # noqa: E402

name, fast = (sys.argv[1].split(":") + [""])[:2]
passes, cur = [], []
for a in sys.argv[2:]:
    if a == "--":
        passes.append(cur)
        cur = []
    else:
        cur.append(a)
if cur:
    passes.append(cur)
# (every pass under its own short timeout: a counter set the profiler cannot schedule has been seen to hang, not fail)
vals, kernel = bench._collect_counters(name, 0, int(os.environ.get("PMC_TIMEOUT", "90")), {"fast": 1, "own": 2}.get(fast, 0), passes)
if vals is None:
    sys.exit(f"{name}: {kernel}")
print(f"{sys.argv[1]}  kernel {kernel}")
for k in sorted(vals):
    print(f"  {k:40s} {vals[k]:.6g}")
cyc = vals.get("GRBM_GUI_ACTIVE", 0) / 8.0
if cyc:
    for k in ("TA_BUSY_avr", "TA_BUSY_max", "TD_TD_BUSY_sum", "TCP_GATE_EN1_sum"):
        if k in vals:
            print(f"  -> {k} / shader cycles of the launch = {vals[k] / cyc:.4f}")
    if "TA_TA_BUSY_sum" in vals:
        print(f"  -> TA_TA_BUSY_sum / (256 CUs x cycles) = {vals['TA_TA_BUSY_sum'] / (256 * cyc):.4f}")
    for k in ("TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_ADDR_STALLED_BY_TD_CYCLES_sum", "TCP_PENDING_STALL_CYCLES_sum",
              "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_TA_TCP_STATE_READ_sum"):
        if k in vals:
            print(f"  -> {k} / (256 CUs x cycles) = {vals[k] / (256 * cyc):.4f}")
