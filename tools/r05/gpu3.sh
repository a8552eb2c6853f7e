#!/bin/bash
# Round 5, GPU session 3: own-tree tests again (tie-aware agreement), A/B of the shipped exact kernels against round 5's
# "tex" development build (did the macro clean-up cost the plane 8 %?).  Output -> gpurun_out/r05_gpu3/
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu3
out=gpurun_out/r05_gpu3; mkdir -p $out
export TMPDIR=/tmp
{
  date
  timeout 1200 python -m pytest -q -s tests/test_gpu_own_tree.py 2>&1 | grep -v "^$" | grep "^\[\|passed\|failed\|Error\|assert" | tail -80
  date
} > $out/tests.txt 2>&1
{
  for spec in "plane 1280 64" "corpus:features1 1280 16" "cornell1m 1024 16"; do
    set -- $spec
    for lib in build/dev/libythip_tex.so yocto-gl_amd/csrc/libythip.so build/dev/libythip_tex.so yocto-gl_amd/csrc/libythip.so; do
      printf "%-40s " $lib
      YTHIP_LIB=$PWD/$lib SCENE=$1 RES=$2 SPP=$3 SAMPLERS=path LAUNCHES=6 DIGEST=1 timeout 300 python tools/sampler_times.py 2>&1 | tail -n 1
    done
  done
  date
} > $out/ab.txt 2>&1
cat $out/tests.txt $out/ab.txt
