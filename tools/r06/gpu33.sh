#!/bin/bash
# round 6, session 33: finer direction codes in the sort key (octant x major axis)
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu33; mkdir -p $out
export TMPDIR=/tmp
{
  date
  SCENES=cfg2b VARIANTS=0:4,3:3,4:3,3:2,4:2,3:4 LAUNCHES=3 timeout 600 python tools/r06/stream_ab.py
  SCENES=cornell9m VARIANTS=0:4,3:3,4:3,3:2 LAUNCHES=3 timeout 600 python tools/r06/stream_ab.py
  SCENES=configs4 VARIANTS=0:3,3:3,4:3,3:2 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=configs3 SPP=64 VARIANTS=1:3,3:3,4:3,4:2 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  date
} > $out/stream_ab_dircodes.txt 2>&1
cat $out/stream_ab_dircodes.txt
