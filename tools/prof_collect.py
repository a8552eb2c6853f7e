#!/usr/bin/env python
"""Condense a tools/prof.sh output directory into the small files committed under
profiles/:  <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats),
<tag>_pmc.json (per-kernel counter sums / per dispatch) and traffic.json (HBM
bytes per k_trace launch = 2 x FETCH_SIZE + WRITE_SIZE, both reported in KiB by
rocprofv3; the factor 2 is the gfx950 FETCH_SIZE correction of
MI355X_MICROARCH.md §HBM)."""
import collections, csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "gpurun_out", "profiles_" + tag)
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "kt", "**", "*_kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
for line in open(os.path.join(src, "kt.log")):
    if line.startswith("{"):
        open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w").write(line)
out = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("yt::"):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
        out.setdefault(k, {})["vgpr"] = int(r["VGPR_Count"])
        out[k]["sgpr"] = int(r["SGPR_Count"])
        out[k]["lds_bytes"] = int(r["LDS_Block_Size"])
        out[k]["scratch_bytes"] = int(r["Scratch_Size"])
    for k in acc:
        n = len(disp[k])
        for c, v in acc[k].items():
            out[k][c] = {"dispatches": n, "sum": v, "per_dispatch": v / n}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
traffic = {}
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        fetch = v["FETCH_SIZE"]["per_dispatch"] * 1024
        write = v["WRITE_SIZE"]["per_dispatch"] * 1024
        traffic[k.replace("yt::", "").split("<")[0]] = {
            "kernel": k, "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
            "hbm_bytes_per_launch": 2 * fetch + write,
            "note": "2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, mean over "
                    f"{v['FETCH_SIZE']['dispatches']} launches of bench.py's step"}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
