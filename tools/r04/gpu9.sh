#!/bin/bash
# Round 4, GPU session 9: the first survivor of a TLAS leaf entered with the ray its pretest transformed (-DYT_DIRECT_ENTER).
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu9; mkdir -p $out
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
  ab plane 1280 64 base6 direct base6 direct
  ab cornell1m 1024 16 base6 direct base6 direct
  ab cfg4 1920 32 base6 direct base6 direct
  ab cfg5 1280 16 base6 direct base6 direct
  ab cornell9m 1024 16 base6 direct
  ab instances 1280 64 base6 direct
  ab lines_points 1280 64 base6 direct
  ab materials 1280 64 base6 direct base6 direct
  ab cornellbox 1280 64 base6 direct
  ab corpus:materials1 1280 64 base6 direct base6 direct
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
