#!/bin/bash
# round 6, session 22: kernel trace of a streamed batch with the tail kernel (where does the batch's time go now?)
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu22; mkdir -p $out
export TMPDIR=/tmp
for sc in ${SCENES:-cfg2b configs4}; do
  v=1:3; [ $sc = configs4 ] && v=2:4
  for fin in ${FINS:-0 250}; do
    rm -rf /tmp/tr; SCENE=$sc VARIANT=$v FINISH=$fin GROUPS=${GROUPS_N:-2} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python tools/r06/gen_profile.py > $out/gen_${sc}_$fin.txt 2>&1
    { echo "== $sc finish $fin"; head -n 1 $out/gen_${sc}_$fin.txt; python tools/r06/trace_sum.py /tmp/tr; } | tee -a $out/trace_sum.txt
  done
done
