#!/bin/bash
# Single-GPU probe of the N-GPU strong-scaling bound: time the slice each rank of
# N would render (bench.py --as-rank R/N), no gather.  Usage: tools/scale_probe.sh OUT [bench args]
out=${1:-gpurun_out/scale_probe.txt}; shift
: > $out
for spec in "columns 0/2" "columns 0/4" "columns 0/8" "columns 5/8" "rows 0/8" "rows 4/8" "rows 7/8"; do
  set -- $spec "${@:3}"
  echo "== $1 $2 ${@:3}" >> $out
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --sharding $1 --as-rank $2 "${@:3}" >> $out 2>&1
done
