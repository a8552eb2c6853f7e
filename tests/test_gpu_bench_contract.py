"""bench.py's output contract (one JSON line, the keys the driver reads, the roofline
and cpu_baseline objects) checked on a short run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


LINE_LIMIT = 4096  # the driver could not parse round 4's 25.8 KB line: the printed line stays under 4 KB


def run_bench(*args, detail=None):
    extra = ["--detail", detail] if detail else []
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, *extra], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "rank 0 prints exactly ONE line on stdout"
    assert len(lines[0]) < LINE_LIMIT, len(lines[0])
    return json.loads(lines[0])


def test_default_line_carries_the_contract(tmp_path):
    import time
    t0 = time.time()
    side = str(tmp_path / "detail.json")
    line = run_bench("--steps", "2", "--warmup", "1", detail=side)
    wall = time.time() - t0
    # the printed line: contract keys + a roofline one can check by hand + the CPU baseline + one short entry per other workload
    for k in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "other_configs", "detail"]:
        assert k in line, k
    assert "truncated" not in line and line["detail"] == side
    assert line["config"]["resolution"] == [1280, 720] and "workload" in line["config"]
    lr = line["roofline"]
    for k in ["kernel", "launch_ms_avg", "fractions", "bound", "frac", "achieved", "peak", "unit", "lane_utilisation",
              "wave_wait_share", "traffic", "algorithmic_bytes_per_launch"]:
        assert k in lr, k
    assert "counters_per_launch" not in lr
    for o in line["other_configs"]:
        assert set(o) <= {"name", "mode", "value", "ms_per_step", "bound", "frac", "lanes", "x"}, o
        assert o["value"] > 0 and o["mode"] in ("bit-exact", "tolerance", "own-tree", "stream", "pathdirect", "pathmis", "own-stream", "direct-stream")
        if o["mode"] == "bit-exact":
            assert o["bound"] in ("hbm", "l2", "valu", "ta") and 0 < o["frac"] <= 1 and 0 < o["lanes"] <= 1
    lb = line["cpu_baseline"]
    assert lb["kind"] == "reference" and lb["nproc"] >= 1 and lb["cores"] >= 1 and lb["cpu_model"]
    assert str(lb["cores"]) in lb["sweep"] and lb["sweep"][str(lb["cores"])] == max(lb["sweep"].values())
    assert wall < 120, wall  # (includes python start-up and the CPU baseline's sweep; the driver saw 81.7 s in round 4)
    print(f"[bench] default run: wall {wall:.1f} s, printed line {len(json.dumps(line, separators=(',', ':')))} bytes")
    # the side file: the full record (the printed line is a projection of it)
    j = json.load(open(side))
    assert j["value"] == line["value"] and j["roofline"]["frac"] == lr["frac"]
    assert len(j["roofline"]["counters_per_launch"]) >= 15
    for k in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]:
        assert k in j, k
    assert j["metric"] == "Msamples/s" and j["unit"] == "Msamples/s" and j["higher_is_better"] is True
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1
    assert j["vs_baseline"] is None and j["data"] == "synthetic" and j["dtype"] == "f32"
    c = j["config"]
    assert "workload" in c and "model" not in c
    assert c["resolution"] == [1280, 720] and c["spp"] == 64 and c["triangles"] == 1000000
    # value = units / time
    assert abs(j["value"] - 1280 * 720 * 64 / j["ms_per_step"] / 1e3) <= 1e-3 * j["value"]
    r = j["roofline"]
    # fractions of the roofs that can bind, from live counters of this workload: all <= 1, the
    # largest is the bound
    assert r["bound"] in ("hbm", "l2", "valu", "ta") and set(r["fractions"]) == {"hbm", "l2", "valu", "ta"}
    assert all(0 < f <= 1 for f in r["fractions"].values()), r["fractions"]
    assert r["frac"] == max(r["fractions"].values()) and r["fractions"][r["bound"]] == r["frac"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    assert r["traffic"] > 0 and abs(r["fractions"]["hbm"] - r["traffic"] / (r["launch_ms_avg"] * 1e-3) / 8e12) < 1e-3
    # the algorithmic byte rate is reported next to it (it exceeds the HBM peak on this cache-resident scene)
    assert abs(r["algorithmic_GBps"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms_avg"] * 1e-3) / 1e9) <= \
        1e-3 * r["algorithmic_GBps"]
    assert 0 < r["launch_ms_avg"] <= j["ms_per_step"] * 1.05  # the kernel is the step
    # the counted work is the reference's (32.59 node pops, 3.35 triangle tests per sample)
    assert abs(r["per_sample"]["nodes"] - 32.592) < 0.01 and abs(r["per_sample"]["triangles"] - 3.348) < 0.01
    b = j["cpu_baseline"]
    if "error" not in b:
        assert b["kind"] == "reference" and b["unit"] == "Msamples/s" and b["cores"] >= 1 and b["value"] > 0
        assert j["value"] > b["value"]
    # cfg2b, configs[3], configs[4] (hair), the cache-exceeding Cornell box and two scenes of the reference's corpus (the
    # general kernel class) bit-exact, then the BASELINE workloads in the tolerance mode — each with its own fractions
    assert isinstance(j.get("other_configs"), list)
    exact = [o for o in j["other_configs"] if o.get("mode", "").startswith("bit-exact")]
    fast = [o for o in j["other_configs"] if o.get("mode", "").startswith("tolerance")]
    assert [o["name"] for o in exact] == ["cfg2b", "configs3", "configs4", "cornell9m", "materials1", "features1"]
    assert [o["name"] for o in fast] == ["configs1", "cfg2b", "configs3", "configs4", "cornell9m", "materials1", "features1"]
    for o in j["other_configs"]:
        assert "error" not in o, o
        assert o["value"] > 0 and o["unit"] == "Msamples/s"
    for o in exact:  # (the tolerance entries' counter passes are opt-in: --tolerance-counters)
        f = o["roofline"]["fractions"]
        assert set(f) == {"hbm", "l2", "valu", "ta"} and all(0 < x <= 1 for x in f.values()), (o["workload"], f)
    assert all(o["roofline"]["kernel"].startswith("yt::k_trace") for o in exact)
    assert all(o["fastmath_ran"] == 1 for o in fast)  # the tolerance-mode unit really ran
    assert all(0.9 < o["speedup_over_bit_exact"] < 2 for o in fast)
    # round 6: `path` on the streaming scheduler (bit-exact) on the workloads it was measured on, and the NEE samplers on cfg2b
    streamed = [o for o in j["other_configs"] if o.get("mode", "").startswith("stream")]
    assert [o["name"] for o in streamed] == ["cfg2b", "configs3", "configs4", "cornell9m"]
    assert all(o["streamed"] == 1 and o["stream"]["groups"] == 2 and o["stream"]["generations"] >= 16 for o in streamed)  # (>= the batch's samples; the tail kernel takes the rest)
    assert all(0.5 < o["speedup_over_bit_exact"] < 2 for o in streamed)
    nee = [o for o in j["other_configs"] if o.get("mode", "").startswith("path")]
    assert [(o["name"], o["mode"].split(":")[0]) for o in nee] == [("cfg2b", "pathdirect"), ("cfg2b", "pathmis")]
    assert all("sampler=" + o["mode"].split(":")[0] in o["workload"] and o["streamed"] == 0 for o in nee)
    ds = [o for o in j["other_configs"] if o.get("mode", "").startswith("direct-stream")]  # (round 6, last session: pathdirect streamed)
    assert [o["name"] for o in ds] == ["cfg2b"] and ds[0]["streamed"] == 1 and "sampler=pathdirect" in ds[0]["workload"]
    assert 0.8 < ds[0]["speedup_over_bit_exact"] < 3
    own = [o for o in j["other_configs"] if o.get("mode", "").startswith("own-tree")]
    assert [o["name"] for o in own] == [o["name"] for o in fast]
    assert all(o["fastmath_ran"] == 2 and o["own_tree"]["nodes"] > 0 for o in own)  # ... and so did the own-tree unit, on its tree
    assert all(0.8 < o["speedup_over_bit_exact"] < 4 for o in own)
    # ... and the own tree on the streaming scheduler (the scheduler's kernels of the own-tree unit), where the scheduler pays
    own_streamed = [o for o in j["other_configs"] if o.get("mode", "").startswith("own-stream")]
    assert [o["name"] for o in own_streamed] == ["cfg2b", "configs3", "cornell9m"]
    assert all(o["fastmath_ran"] == 2 and o["streamed"] == 1 and o["stream"]["groups"] == 2 for o in own_streamed)
    assert all(0.8 < o["speedup_over_bit_exact"] < 4 for o in own_streamed)
    assert "800,000 line segments" in exact[2]["workload"]
    assert exact[4]["roofline"]["kernel"].endswith("3>") and exact[5]["roofline"]["kernel"].endswith("0>")  # opaque-textured / general class
    big = exact[3]
    assert big["baked_bvh_bytes"] > 400e6  # larger than the 256 MB Infinity Cache


def test_slice_run_and_flags():
    j = run_bench("--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline", "--as-rank", "3/8")
    assert j["config"]["pixels_per_rank"] == 1280 * 720 // 8 and "roofline" not in j and "cpu_baseline" not in j
    assert abs(j["value"] - j["config"]["pixels_per_rank"] * 64 / j["ms_per_step"] / 1e3) <= 1e-3 * j["value"]


def test_gather_path_over_rccl_with_one_rank():
    """The N > 1 code path of bench.py on the backend the driver's launch uses — init_process_group("nccl") = RCCL, the frame
    gathered by all_gather on device tensors, the barrier, the max-over-ranks all_reduce — with the one rank a 1-GPU box has
    (--rehearse-gather).  What it cannot show is a transfer between two devices; what it does show is that RCCL comes up in this
    image and that the collective runs on the kernel's stream."""
    j = run_bench("--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-other-configs", "--rehearse-gather")
    assert j["n_gpus"] == 1 and j["config"]["collective"] == {"backend": "nccl", "ranks": 1}
    assert abs(j["value"] - 1280 * 720 * 64 / j["ms_per_step"] / 1e3) <= 1e-3 * j["value"]
    assert j["value"] > 5000  # (the gather of one rank is a device copy: the step stays the kernel)


def test_two_rank_launch_rehearsed_with_gloo():
    """The N > 1 path of bench.py launched exactly as the driver launches it
    (torch.distributed.run, one process per rank) — rehearsed on this one GPU with the
    gloo backend (YTHIP_DIST_BACKEND; both ranks share the device, so the timings mean
    nothing): BASELINE configs[2] (the frame split over the ranks) as the primary line,
    column sharding, framebuffer gather, MAX over ranks, the weak-scaling leg, one JSON
    line from rank 0."""
    env = dict(os.environ, YTHIP_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29700 + os.getpid() % 200
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--resolution", "320", "--spp", "4"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert len(lines[0]) < LINE_LIMIT, len(lines[0])
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["steps"] == 2
    w, h = j["config"]["resolution"]
    assert (w, h) == (320, 180) and j["config"]["pixels_per_rank"] == w * h // 2  # the SAME frame, split
    assert "configs[2]" in j["config"]["workload"] and j["config"]["sharding"] == "columns/2"
    assert j["config"]["collective"] == {"backend": "gloo", "ranks": 2}
    # a run explains its own efficiency: per rank, ms per step in its slice's kernel and in the gather that follows
    pr = j["config"]["per_rank"]
    assert len(pr["slice_ms"]) == 2 and len(pr["gather_ms"]) == 2 and all(t > 0 for t in pr["slice_ms"] + pr["gather_ms"])
    assert max(a + b for a, b in zip(pr["slice_ms"], pr["gather_ms"])) <= j["ms_per_step"] * 1.5 + 1.0
    assert abs(j["value"] - w * h * 4 / j["ms_per_step"] / 1e3) <= 1e-3 * j["value"]
    s = j["weak_scaling"]
    assert s["resolution"] == [448, 252] and s["pixels_per_rank"] == 448 * 252 // 2 and s["value"] > 0  # weak_resolution(320, 2)
    assert len(s["per_rank"]["slice_ms"]) == 2
    assert "cpu_baseline" not in j and "other_configs" not in j  # rank 0 at N=1 only


def test_eight_rank_launch_rehearsed_with_gloo():
    """`bench.py --gpus 8` as the driver launches it at round end: eight processes under
    torch.distributed.run, rehearsed on this ONE GPU with gloo (all ranks share the device: the timings mean nothing,
    the plumbing is what is tested) — every rank renders its 1/8 of the tile columns, the frame is gathered, rank 0 prints
    ONE line under the size limit with eight per-rank slice / gather times, the strong-scaling value and the weak-scaling leg."""
    env = dict(os.environ, YTHIP_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29400 + os.getpid() % 200
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "8", "--steps", "2", "--warmup", "1", "--resolution", "640", "--spp", "4"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert len(lines[0]) < LINE_LIMIT, len(lines[0])
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and j["steps"] == 2 and j["warmup"] == 1
    w, h = j["config"]["resolution"]
    assert (w, h) == (640, 360) and j["config"]["pixels_per_rank"] == w * h // 8  # 40 tile columns: 5 per rank
    assert "configs[2]" in j["config"]["workload"] and j["config"]["sharding"] == "columns/8"
    assert j["config"]["collective"] == {"backend": "gloo", "ranks": 8}
    pr = j["config"]["per_rank"]
    assert len(pr["slice_ms"]) == 8 and len(pr["gather_ms"]) == 8 and all(t > 0 for t in pr["slice_ms"] + pr["gather_ms"])
    assert abs(j["value"] - w * h * 4 / j["ms_per_step"] / 1e3) <= 1e-3 * j["value"]
    s = j["weak_scaling"]
    assert s["pixels_per_rank"] * 8 == s["resolution"][0] * s["resolution"][1] and s["value"] > 0
    assert len(s["per_rank"]["slice_ms"]) == 8 and len(s["per_rank"]["gather_ms"]) == 8
    assert "cpu_baseline" not in j and "other_configs" not in j
