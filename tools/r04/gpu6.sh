#!/bin/bash
# Round 4, GPU session 6: the wide step in two 64-B halves (YT_HALF_STEP) at 4 and at 5 waves per SIMD.  -> gpurun_out/r04_gpu6/
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu6; mkdir -p $out
export TMPDIR=/tmp
ab() {
  local sc=$1 res=$2 spp=$3; shift 3
  for n in "$@"; do
    printf "%-10s " $n
    YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=path DIGEST=1 timeout 90 python tools/sampler_times.py 2>&1 | grep -v "^\[timing\]" | tail -n 1
  done
}
{
  date
  ab plane 1280 64 base4 half w5 half5
  ab cornell1m 1024 16 base4 half w5 half5
  ab cfg4 1920 32 base4 half w5 half5
  ab cfg5 1280 16 base4 half w5 half5
  ab cornell9m 1024 16 base4 half w5 half5
  ab materials 1280 64 base4 half w5 half5
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
