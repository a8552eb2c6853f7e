#!/usr/bin/env python
"""Sums a rocprofv3 kernel trace (csv) by kernel: launches, total / mean / max duration, and the span of the LAST streamed batch
(from its ks_init to its last kernel) with every kernel's share of it.   python tools/r06/trace_sum.py DIR"""
import csv, glob, re, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
inits = [k for k, r in enumerate(rows) if "ks_init" in r["Kernel_Name"]]
groups = 1
if len(inits) > 1 and inits[-1] - inits[-2] < 4: groups = 2
first = inits[-groups] if inits else 0
batch = rows[first:]
t0, t1 = int(batch[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in batch)
print(f"last batch: {len(batch)} launches, span {(t1 - t0) / 1e6:.3f} ms")
acc = defaultdict(lambda: [0, 0, 0])
for r in batch:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = acc[name]; a[0] += 1; a[1] += d; a[2] = max(a[2], d)
for name, (n, tot, mx) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:70]:70s} {n:6d} launches {tot / 1e6:9.3f} ms  mean {tot / n / 1e3:8.1f} us  max {mx / 1e3:9.1f} us")
# busy time (union of kernel intervals) vs span: how much of the batch has no kernel running at all
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in batch)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"some kernel running: {busy / 1e6:.3f} ms of {(t1 - t0) / 1e6:.3f} ms")
