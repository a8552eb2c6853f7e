#!/bin/bash
# Round 5, GPU session 12 (first of the re-entered session): the whole GPU suite on HEAD with the own-tree / tolerance gates
# printed (-s), then the round's evidence (tools/prof_r05.sh, its own-tree step skipped: the suite run above printed the gates).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu12
out=gpurun_out/r05_gpu12; mkdir -p $out
export TMPDIR=/tmp
{ date; timeout 1500 python -m pytest tests -m gpu -x -q -s > $out/suite_full.log 2>&1; echo "rc=$?"; tail -3 $out/suite_full.log; date; } > $out/suite.txt 2>&1
grep "hits\]\|gate\]\|fastmath\]" $out/suite_full.log | sed 's/^\.*//' > $out/gates.txt
SKIP_OWN_GATES=1 bash tools/prof_r05.sh r05 > $out/prof.log 2>&1
cat $out/suite.txt; tail -3 $out/prof.log; cat gpurun_out/profiles_r05/r05_bench_wall.txt; cat gpurun_out/profiles_r05/r05_bench_line_default.json
