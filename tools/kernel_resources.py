#!/usr/bin/env python
"""Registers, spills, scratch, LDS and occupancy of every kernel of libythip's k_trace units + ythip.hip under the shipped flags:
compiles the unit with -Rpass-analysis=kernel-resource-usage (device side only, no object kept) and tabulates the
remarks.   python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt   (a few minutes of hipcc, no GPU)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402  (the shipped flags)

# (round 4: the kernels live in several units — the three k_trace units, the tolerance-mode unit, the C-ABI unit)
UNITS = ["yt_trace_path.hip", "yt_trace_nee.hip", "yt_trace_nee_cls.hip", "yt_trace_misc.hip", "yt_stream.hip", "yt_fast.hip", "yt_owntree.hip", "ythip.hip"]
if len(sys.argv) > 1:  # remarks captured earlier (hipcc ... 2> file)
    remarks = open(sys.argv[1]).read()
else:
    from concurrent.futures import ThreadPoolExecutor

    def one(unit):
        src = os.path.join(ROOT, "yocto-gl_amd", "csrc", unit)
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + G.HIPCC_FLAGS + ["--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c",
                            "-o", "/dev/null", src], capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
        return r.stderr

    with ThreadPoolExecutor(max_workers=len(UNITS)) as pool:
        remarks = "\n".join(pool.map(one, UNITS))
rows, cur = [], None
for line in remarks.splitlines():
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
names = subprocess.run(["c++filt"], input="\n".join(x["name"] for x in rows), capture_output=True, text=True).stdout.splitlines()
print(f"# hipcc {' '.join(G.HIPCC_FLAGS)} -Rpass-analysis=kernel-resource-usage on csrc/" + "{yt_trace_path,yt_trace_nee,yt_trace_misc,yt_fast,yt_owntree,ythip}.hip (tools/kernel_resources.py).")
print("# k_trace<SAMPLER, LP, COUNT, WIDE, CLS>: SAMPLER = ythip_sampler (0 path, 1 pathdirect, 2 pathmis, 3 pathtest, 4 naive ... 8 falsecolor); LP 0 no area")
print("# lights / 2 walk stage; COUNT = work-counting launch; WIDE = wide (quad-record) walk; CLS 0 general, 1 matte + triangles + no textures,")
print("# 2 no textures, 3 opaque textured (matte / glossy / reflective, colour + normal textures, triangles + quads).  yt_fast:: = the tolerance-mode unit, yt_own:: = the own-tree unit.")
print("# Spills are registers (VGPR spills live in scratch, SGPR spills in VGPR lanes).")
print(f"{'kernel':66s} {'VGPR':>5s} {'SGPR':>5s} {'VGPRspill':>9s} {'SGPRspill':>9s} {'scratchB':>8s} {'LDS':>6s} {'occ':>5s}")
for row, name in zip(rows, names):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(yt(_fast|_own)?::DScene, yt(_fast|_own)?::DState, yt(_fast|_own)?::KParams\)$", "", name)
    print(f"{name[:66]:66s} {row.get('VGPRs', '?'):>5s} {row.get('TotalSGPRs', row.get('SGPRs', '?')):>5s} {row.get('VGPRs Spill', '?'):>9s} {row.get('SGPRs Spill', '?'):>9s} "
          f"{row.get('ScratchSize', '?'):>8s} {row.get('LDS Size', '?'):>6s} {row.get('Occupancy', '?'):>5s}")
