#!/bin/bash
# Round 5, GPU session 16: the shipped build — random-scene parity (tools/fuzz_parity.py, 300 fresh seeds), the single-GPU probe of
# the N-GPU strong-scaling bound (bench.py --as-rank), then the whole GPU suite with the gates printed and the round's evidence
# (tools/prof_r05.sh, its own-tree step skipped: the suite run prints the gates).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu16
out=gpurun_out/r05_gpu16; mkdir -p $out
export TMPDIR=/tmp
{ date; timeout 400 python tools/fuzz_parity.py 7000 300 2>&1 | tail -n 320; date; } > $out/fuzz.txt 2>&1
{
  for spec in "0/1" "0/2" "0/4" "0/8" "5/8"; do
    echo "== configs[1] columns $spec"
    timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --sharding columns --as-rank $spec 2>/dev/null | tail -n 1 | cut -c1-400
  done
} > $out/scale_probe.txt 2>&1
{ date; timeout 1500 python -m pytest tests -m gpu -x -q -s > $out/suite_full.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $out/suite_full.log | tail -1; date; } > $out/suite.txt 2>&1
grep "hits\]\|gate\]\|fastmath\]" $out/suite_full.log | sed 's/^\.*//' > $out/gates.txt
SKIP_OWN_GATES=1 bash tools/prof_r05.sh r05 > $out/prof.log 2>&1
tail -3 $out/fuzz.txt; cat $out/scale_probe.txt | cut -c1-200; cat $out/suite.txt; cat gpurun_out/profiles_r05/r05_bench_wall.txt; cat gpurun_out/profiles_r05/r05_bench_line_default.json
