#!/bin/bash
# Round 4, GPU session 14: two small experiments against the final build's twin (base8) —
#   nt     -DYT_NT_STATE   per-pixel accumulators read / written with the non-temporal hint
#   idhit  -DYT_ID_ON_HIT  a line's element id (third float4 of its leaf record) fetched only for a hit
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu14; mkdir -p $out
export TMPDIR=/tmp LAUNCHES=6
ab() {
  local sc=$1 res=$2 spp=$3; shift 3
  for n in "$@"; do
    printf "%-10s " $n
    YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=${SAMPLERS:-path} DIGEST=1 timeout 90 python tools/sampler_times.py 2>&1 | grep -v "^\[timing\]" | tail -n 1
  done
}
{
  date
  ab plane 1280 64 base8 nt base8 nt
  ab cornell1m 1024 16 base8 nt base8 nt
  ab cfg4 1920 16 base8 nt base8 nt
  ab cfg5 1280 16 base8 nt idhit base8 nt idhit
  ab cornell9m 1024 16 base8 nt
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
