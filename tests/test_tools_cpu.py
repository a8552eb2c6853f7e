"""CPU checks of the round-2 helpers that run around the GPU path (no device needed)."""
import os
import sys

import numpy as np

import parity as P

ROOT = P.ROOT
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd"))


def test_fuzz_scenes_are_reproducible_and_always_lit():
    """tools/fuzz_parity.py: a seed names one scene + one parameter set, and every scene has a
    light the reference's make_trace_lights would find (the reference reads out of bounds in
    sample_lights otherwise)."""
    import fuzz_parity as F
    samplers, streamed = set(), 0
    for seed in range(60):
        a, pa, ha, sa = F.random_scene(seed)
        b, pb, hb, sb = F.random_scene(seed)
        assert pa == pb and ha == hb and sa == sb
        if sa:  # (round 6) a case for the streaming scheduler: `path` / `pathdirect`, real batches
            streamed += 1
            assert pa["sampler"] in ("path", "pathdirect", "naive", "pathtest") and pa["batch"] >= 4 and pa["samples"] % pa["batch"] == 0
        assert a.positions.tobytes() == b.positions.tobytes() and a.materials.tobytes() == b.materials.tobytes()
        samplers.add(pa["sampler"])
        lit = len(a.environments) > 0 or any(
            np.any(a.materials["emission"][i["material"]] > 0) and
            (a.shapes[i["shape"]]["num_triangles"] > 0 or a.shapes[i["shape"]]["num_quads"] > 0) for i in a.instances)
        assert lit, seed
    assert len(samplers) >= 6 and streamed >= 2


def test_kernel_resources_tool_parses_compiler_remarks(tmp_path):
    """tools/kernel_resources.py on captured -Rpass-analysis=kernel-resource-usage remarks (no hipcc run)."""
    import subprocess
    import sys
    tag = "[-Rpass-analysis=kernel-resource-usage]"
    where = "yocto-gl_amd/csrc/yt_kernels.h:1049:1: remark:"
    lines = [f"{where} Function Name: _ZN2yt7k_traceILi0ELi0ELb0ELb1ELi1EEEvNS_6DSceneENS_6DStateENS_7KParamsE {tag}",
             " 1049 |     k_trace(DScene sc, DState st, KParams kp) {", "      | ^"]
    for key, value in [("TotalSGPRs", 104), ("VGPRs", 128), ("AGPRs", 0), ("ScratchSize [bytes/lane]", 976), ("Dynamic Stack", "False"),
                       ("Occupancy [waves/SIMD]", 4), ("SGPRs Spill", 157), ("VGPRs Spill", 2), ("LDS Size [bytes/block]", 10016)]:
        lines.append(f"{where}     {key}: {value} {tag}")
    remarks = tmp_path / "remarks.txt"
    remarks.write_text("\n".join(lines) + "\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_resources.py"), str(remarks)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    row = [l for l in r.stdout.splitlines() if l.startswith("yt::k_trace<0, 0, false, true, 1>")]
    assert len(row) == 1 and row[0].split()[-7:] == ["128", "104", "2", "157", "976", "10016", "4"], r.stdout


def test_bench_table_tool_renders_the_committed_line():
    """tools/bench_table.py on profiles/r06_bench_detail.json (the side file of the committed line): DESIGN.md §5's table is this
    output, so the table and the bench line are one run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = os.path.join(root, "profiles", "r06_bench_detail.json")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_table.py"), line, os.path.join(root, "profiles", "r05_bench_detail.json")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l for l in r.stdout.splitlines() if l.startswith("| ")]
    # header, seven workloads bit-exact, in the tolerance mode, on the own tree, four streamed, two NEE samplers, three own-tree streamed, pathdirect streamed
    # (the rule row starts "|-")
    assert len(rows) == 1 + 7 + 7 + 7 + 4 + 2 + 3 + 1
    import json
    j = json.load(open(line))
    assert f"**{j['value']:,.0f}**" in rows[1] and "configs[1]" in rows[1]
    design = open(os.path.join(root, "DESIGN.md")).read()
    assert rows[1] in design and rows[-1] in design  # DESIGN.md carries exactly this table


def test_the_round5_kit_still_compiles():
    """The macro-guarded leads of DESIGN.md §7e (tools/r05_first_session.sh) are off in the shipped build; this keeps them
    compiling — a syntax-only device pass of the units that see them, a few seconds each."""
    import shutil
    import subprocess
    import pytest
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    kit = ["-DYT_WHOLE_RECORD=1", "-DYT_RECORDS_BY_VALUE", "-DYT_LIBM_NO_TABLES", "-DYT_SRGB_LUT", "-DYT_TEXELS_TOGETHER",
           "-DYT_SURFACE_BY_VALUE"]
    for unit in ["ythip.hip", "yt_trace_path.hip", "yt_fast.hip"]:
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-fsyntax-only", *kit,
                            os.path.join(ROOT, "yocto-gl_amd", "csrc", unit)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, unit + "\n" + r.stderr[-3000:]


def test_bench_line_stays_under_4k_whatever_the_detail_holds():
    """Round 4's 25.8 KB bench line could not be parsed by the driver.  compact_line() projects the full record onto
    a line under 4 KB — checked here on that very record, and on an inflated 8-rank record with three modes per workload."""
    import copy
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    full = json.load(open(os.path.join(root, "profiles", "r04_bench.json")))
    assert len(json.dumps(full)) > 20000
    text = bench.compact_line(full, "bench_detail.json")
    assert len(text) < bench.LINE_LIMIT == 4096 and "\n" not in text
    line = json.loads(text)
    assert line["value"] == full["value"] and line["roofline"]["frac"] == full["roofline"]["frac"]
    assert "counters_per_launch" not in line["roofline"] and "truncated" not in line
    assert [o["name"] for o in line["other_configs"]] == [o["name"] for o in full["other_configs"]]
    assert all(set(o) <= {"name", "mode", "value", "ms_per_step", "bound", "frac", "lanes", "x"} for o in line["other_configs"])
    # N = 8, both scaling legs with per-rank times, and a third mode per workload: still one line under the limit
    big = copy.deepcopy(full)
    big["n_gpus"] = 8
    per_rank = {"slice_ms": [1.23456] * 8, "gather_ms": [0.12345] * 8}
    big["config"].update(collective={"backend": "nccl", "ranks": 8}, per_rank=per_rank, sharding="columns/8",
                         framebuffer_gather="rccl all_gather + un-permute per step, on the kernel stream")
    big["weak_scaling"] = {"value": 1.0, "unit": "Msamples/s", "ms_per_step": 1.0, "scaling": "weak", "resolution": [3584, 2016],
                           "spp": 64, "pixels_per_rank": 903168, "per_rank": per_rank, "note": "x" * 500}
    big["other_configs"] = big["other_configs"] + [dict(o, mode="own-tree (fastmath = 2)") for o in big["other_configs"][:7]]
    big["cpu_baseline"].update(host={"nproc": 256, "cgroup_cpu_max": "max 100000", "cpu_model": "AMD EPYC 9575F 64-Core Processor"},
                               sweep=[{"cpus": k, "Msamples_per_s": 12.345} for k in (1, 16, 64, 128, 256)])
    text = bench.compact_line(big, "bench_detail.json")
    assert len(text) < 4096
    line = json.loads(text)
    for k in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "detail"]:
        assert k in line, k


def test_check_bench_line_recomputes_the_roofline_from_its_side_files():
    """tools/check_bench_line.py on round 5's committed evidence: the line, its side file of raw counters and the
    kernel trace of the same session agree; a line whose fraction, traffic or launch time was edited does not pass."""
    import copy
    import csv
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import check_bench_line as cbl
    prof = os.path.join(root, "profiles")
    line = json.loads(open(os.path.join(prof, "r05_bench.json")).read())
    detail = json.load(open(os.path.join(prof, "r05_bench_detail.json")))
    rows = list(csv.DictReader(open(os.path.join(prof, "r05_kernel_stats.csv"))))
    bad, out = cbl.check(line, detail, rows)
    assert bad == [] and abs(out["fractions"]["valu"] - line["roofline"]["frac"]) < 0.01 and 2.0 < out["shader_clock_GHz"] < 2.6
    assert cbl.main(["x", os.path.join(prof, "r05_bench.json"), os.path.join(prof, "r05_bench_detail.json"),
                     os.path.join(prof, "r05_kernel_stats.csv")]) == 0
    for edit in (lambda l: l["roofline"].__setitem__("frac", l["roofline"]["frac"] * 1.05),
                 lambda l: l["roofline"]["fractions"].__setitem__("hbm", 0.4),
                 lambda l: l["roofline"].__setitem__("traffic", l["roofline"]["traffic"] * 2),
                 lambda l: l.__setitem__("value", l["value"] * 1.1),
                 lambda l: l["roofline"].__setitem__("launch_ms_avg", l["roofline"]["launch_ms_avg"] * 0.9)):
        forged = copy.deepcopy(line)
        edit(forged)
        assert cbl.check(forged, detail, rows)[0], "an edited line must not pass"
