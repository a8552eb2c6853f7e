#!/bin/bash
# Round 4, GPU session 12: the wavefront-uniform node step in the majority-phase walk (closed-room kernels).
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu12; mkdir -p $out
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
  ab cornell1m 1024 16 base7 phsc base7 phsc
  ab cornell9m 1024 16 base7 phsc base7 phsc
  ab cornellbox 1280 64 base7 phsc base7 phsc
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
