#!/bin/bash
# Round 4, GPU session 7: wavefront-uniform wide steps through the scalar cache (-DYT_SCALAR_STEP).  -> gpurun_out/r04_gpu7/
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu7; mkdir -p $out
export TMPDIR=/tmp
ab() {
  local sc=$1 res=$2 spp=$3; shift 3
  for n in "$@"; do
    printf "%-10s " $n
    YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=${SAMPLERS:-path} DIGEST=1 timeout 90 python tools/sampler_times.py 2>&1 | grep -v "^\[timing\]" | tail -n 1
  done
}
{
  date
  ab plane 1280 64 base5 scalar base5 scalar
  ab cornell1m 1024 16 base5 scalar
  ab cfg4 1920 32 base5 scalar
  ab cfg5 1280 16 base5 scalar
  ab cornell9m 1024 16 base5 scalar
  ab materials 1280 64 base5 scalar
  ab cornellbox 1280 64 base5 scalar
  ab corpus:materials1 1280 64 base5 scalar
  SAMPLERS=eyelight ab plane 1280 64 base5 scalar
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
