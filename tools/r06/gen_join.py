#!/usr/bin/env python
"""Joins tools/r06/gen_profile.py's queue lengths with the kernel trace of the same run: per generation of the LAST streamed batch
the extend / shade / sort times; prints totals by queue-length class."""
import csv, glob, sys
import numpy as np
d, gen = sys.argv[1], np.load(sys.argv[2])
rows = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ext = [(a, b) for a, b, n in rows if "ks_extend" in n]
shd = [(a, b) for a, b, n in rows if "ks_shade" in n]
n = len(gen)
# the last batch's generations are the last ones that did work: take the last `launched` extend launches and align by count
ext, shd = ext[-len(ext) // 2:], shd[-len(shd) // 2:]  # (two streamed batches: warm-up + the measured one)
te = np.array([b - a for a, b in ext[:n]]) / 1e3
ts = np.array([b - a for a, b in shd[:n]]) / 1e3
span = (shd[n - 1][1] - ext[0][0]) / 1e6
print(f"{n} generations, extend {te.sum() / 1e3:.2f} ms, shade {ts.sum() / 1e3:.2f} ms, span {span:.2f} ms")
slots = gen.max()
for lo, hi in ((0.75, 2), (0.5, 0.75), (0.25, 0.5), (0.1, 0.25), (0.0, 0.1)):
    m = (gen[:len(te)] > lo * slots) & (gen[:len(te)] <= hi * slots)
    if m.any():
        r = gen[:len(te)][m].sum()
        print(f"queue in ({lo:.2f}, {hi:.2f}] x slots: {m.sum():4d} generations, {r / 1e6:7.1f} M rays, extend {te[m].sum() / 1e3:7.2f} ms ({r / te[m].sum() / 1e3:6.2f} Grays/s), "
              f"shade {ts[m].sum() / 1e3:7.2f} ms ({ts[m].mean():6.1f} us each)")
