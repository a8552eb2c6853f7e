#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel:
sum and per-dispatch mean of every counter.  Usage: pmc_summary.py DIR [DIR...]"""
import csv, glob, sys, collections, json, os
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        for k in acc:
            if not k.startswith("yt::"): continue
            n = len(disp[k])
            o = out.setdefault(k, {"dispatches": n})
            for c, v in acc[k].items():
                o[c] = {"sum": v, "per_dispatch": v / n}
print(json.dumps(out, indent=1))
