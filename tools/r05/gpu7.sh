#!/bin/bash
# Round 5, GPU session 7: phase split (-DYT_TIMING) of the exact and the own-tree kernels.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu7
out=gpurun_out/r05_gpu7; mkdir -p $out
export TMPDIR=/tmp
{
  for spec in "cornell1m 1024 16" "cfg4 1920 16" "cfg5 1280 16" "corpus:features1 1280 16" "plane 1280 64"; do
    set -- $spec
    for lib in ownT ownO; do for fm in 0 2; do
      echo "== $lib fastmath=$fm $1"
      FASTMATH=$fm YTHIP_LIB=$PWD/build/dev/libythip_$lib.so SCENE=$1 RES=$2 SPP=$3 SAMPLERS=path LAUNCHES=1 timeout 120 python tools/sampler_times.py 2>&1 | grep -v "^\[own" | tail -n 5
    done; done
  done
} > $out/timing.txt 2>&1
cat $out/timing.txt
