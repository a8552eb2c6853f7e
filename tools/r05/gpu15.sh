#!/bin/bash
# Round 5, GPU session 15: the own tree's line shapes built over pieces of their segments (YTHIP_OWN_LINE_SPLIT = shift: 1, 2, 4, 8
# pieces per segment) on the hair (configs[4]); then the hair's hit agreement + gates with the default (4 pieces).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu15
out=gpurun_out/r05_gpu15; mkdir -p $out
export TMPDIR=/tmp
{
  date
  for sh in 0 1 2 3 0 2; do
    printf "split %d  " $sh
    YTHIP_OWN_LINE_SPLIT=$sh FASTMATH=2 YTHIP_LIB=$PWD/build/dev/libythip_ownS.so SCENE=cfg5 RES=1280 SPP=16 SAMPLERS=path LAUNCHES=5 timeout 180 python tools/sampler_times.py 2>&1 | grep -v "^\[" | tail -n 1
  done
  for sh in 0 2; do
    printf "split %d  " $sh
    YTHIP_OWN_LINE_SPLIT=$sh FASTMATH=2 YTHIP_LIB=$PWD/build/dev/libythip_ownS.so SCENE=lines_points RES=1024 SPP=16 SAMPLERS=path LAUNCHES=5 timeout 180 python tools/sampler_times.py 2>&1 | grep -v "^\[" | tail -n 1
  done
  date
  YTHIP_LIB=$PWD/build/dev/libythip_ownS.so timeout 900 python -m pytest -q -s tests/test_gpu_own_tree.py -k "cfg5 or lines_points" 2>&1 | grep "hits\]\|gate\]\|passed\|failed\|Error" | sed 's/^\.*//'
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
