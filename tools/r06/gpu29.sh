#!/bin/bash
# round 6, session 29: the majority-phase walk ON in ks_extend on the uneven workloads
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu29; mkdir -p $out
export TMPDIR=/tmp
{
  date
  SCENES=configs4 VARIANTS=0:3:0,0:3:1 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=configs3 SPP=64 VARIANTS=1:3:0,1:3:1 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=features1,materials1 VARIANTS=2:4:0,2:4:1 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  date
} > $out/stream_ab_phased.txt 2>&1
cat $out/stream_ab_phased.txt
