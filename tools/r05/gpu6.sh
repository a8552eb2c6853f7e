#!/bin/bash
# Round 5, GPU session 6: own-tree variants — A phased walk on the closed rooms only (as the exact kernels), B phased everywhere,
# C children by entry distance, D while-while everywhere.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu6
out=gpurun_out/r05_gpu6; mkdir -p $out
export TMPDIR=/tmp
ab() {
  local sc=$1 res=$2 spp=$3; shift 3
  for n in "$@"; do
    printf "%-8s " $n
    FASTMATH=2 YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=path LAUNCHES=5 timeout 120 python tools/sampler_times.py 2>&1 | grep -v "^\[" | tail -n 1
  done
}
{
  date
  ab cornell1m 1024 16 ownA ownB ownC ownD ownA ownD
  ab cfg4 1920 16 ownA ownB ownC
  ab cfg5 1280 16 ownA ownB ownC
  ab cornell9m 1024 16 ownA ownC ownD
  ab corpus:materials1 1280 16 ownA ownB ownC
  ab corpus:features1 1280 16 ownA ownB ownC
  ab plane 1280 64 ownA ownB ownC
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
