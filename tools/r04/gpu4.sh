#!/bin/bash
# Round 4, GPU session 4: the whole `-m gpu` suite on the adopted build, then the bench line + kernel traces (tools/prof_r04.sh).
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu4; mkdir -p $out
export TMPDIR=/tmp
date > $out/tests.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 >> $out/tests.txt
date >> $out/tests.txt
timeout 1500 bash tools/prof_r04.sh r04 > $out/prof.txt 2>&1
date >> $out/prof.txt
cat $out/tests.txt; tail -5 $out/prof.txt; head -c 3000 gpurun_out/profiles_r04/r04_bench.json
