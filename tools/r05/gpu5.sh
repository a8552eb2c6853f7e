#!/bin/bash
# Round 5, GPU session 5: the instruction budgets (configs[1] and cfg2b), exact / tolerance / own tree.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu5
out=gpurun_out/r05_gpu5; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python tools/instruction_budget.py plane 1280 64 > $out/budget_plane.txt 2>&1
timeout 600 python tools/instruction_budget.py cornell1m 1024 16 > $out/budget_cornell1m.txt 2>&1
cat $out/budget_plane.txt $out/budget_cornell1m.txt
