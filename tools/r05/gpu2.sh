#!/bin/bash
# Round 5, GPU session 2: the own-tree mode's first run — its tests (hit agreement, gates against oracle/_ref), then
# exact / tolerance / own-tree times per workload.  Output -> gpurun_out/r05_gpu2/
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu2
out=gpurun_out/r05_gpu2; mkdir -p $out
export TMPDIR=/tmp
{
  date
  timeout 900 python -m pytest -x -q -s tests/test_gpu_own_tree.py 2>&1 | grep -v "^$" | tail -60
  date
} > $out/tests.txt 2>&1
{
  for spec in "plane 1280 64" "cornell1m 1024 16" "cfg4 1920 32" "cfg5 1280 16" "cornell9m 1024 16" "corpus:materials1 1280 16" "corpus:features1 1280 16"; do
    set -- $spec
    for fm in 0 1 2; do
      printf "fastmath=%d " $fm
      FASTMATH=$fm SCENE=$1 RES=$2 SPP=$3 SAMPLERS=path LAUNCHES=4 timeout 300 python tools/sampler_times.py 2>&1 | tail -n 2 | tr '\n' ' '; echo
    done
  done
  date
} > $out/times.txt 2>&1
cat $out/tests.txt $out/times.txt
