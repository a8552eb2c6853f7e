#!/bin/bash
# Round 4, GPU session 8: wavefront-uniform wide steps through the scalar cache (-DYT_SCALAR_STEP).  -> gpurun_out/r04_gpu8/
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu8; mkdir -p $out
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
  ab plane 1280 64 noscal scalar scal2 scal3
  ab cornell1m 1024 16 noscal scalar scal2 scal3
  ab cfg4 1920 32 noscal scalar scal2 scal3
  ab cfg5 1280 16 noscal scalar scal2 scal3
  ab cornell9m 1024 16 noscal scalar scal2 scal3
  ab materials 1280 64 noscal scalar scal2 scal3
  ab cornellbox 1280 64 noscal scalar scal2 scal3
  ab corpus:materials1 1280 64 noscal scalar scal2 scal3
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
