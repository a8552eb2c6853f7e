#!/bin/bash
# round 6, session 37: more chains with more hardware queues (GPU_MAX_HW_QUEUES: HIP's default is 4)
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu37; mkdir -p $out
export TMPDIR=/tmp
{
  date
  for q in 4 8 16; do
    echo "--- GPU_MAX_HW_QUEUES=$q"
    GPU_MAX_HW_QUEUES=$q SCENES=features1 VARIANTS=2:4:-1:2:0,2:4:-1:4:0,2:4:-1:6:0,2:4:-1:8:0 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
    GPU_MAX_HW_QUEUES=$q SCENES=cfg2b VARIANTS=0:4:-1:2,0:4:-1:3,0:4:-1:4,0:4:-1:6 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  done
  date
} > $out/stream_ab_hwqueues.txt 2>&1
cat $out/stream_ab_hwqueues.txt
