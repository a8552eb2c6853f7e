#!/usr/bin/env python
"""Per hardware queue (= per chain of the streaming scheduler): launches and total kernel time of ks_extend / ks_shade in a rocprofv3
kernel trace — how evenly the two chains share the batch.   python tools/r06/trace_groups.py DIR"""
import csv, glob, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    k = "extend" if "ks_extend" in n else "shade" if "ks_shade" in n else "finish" if "ks_finish" in n else None
    if k:
        a = acc[(r.get("Queue_Id", "?"), k)]
        a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for (q, k), (n, ms) in sorted(acc.items()):
    print(f"queue {q:>4s} {k:7s} {n:5d} launches {ms:9.3f} ms")
