#!/bin/bash
# Round 5, GPU session 27: waves per SIMD for the classes that spill (general class: 259 spilled VGPRs at 4 waves = 128 registers):
# __launch_bounds__(64, 3) = 170 registers, (64, 2) = 256.  Rounds 1-3 measured this on the matte classes only.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu27
out=gpurun_out/r05_gpu27; mkdir -p $out
export TMPDIR=/tmp
ab() {
  local fm=$1 sc=$2 res=$3 spp=$4; shift 4
  for n in "$@"; do
    printf "%-4s fastmath %s " $n $fm
    DIGEST=1 FASTMATH=$fm YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=path LAUNCHES=5 timeout 120 python tools/sampler_times.py 2>&1 | grep -v "^\[" | tail -n 1
  done
}
{
  date
  ab 0 corpus:features1 1280 64 w4 w3 w2
  ab 0 corpus:materials1 1280 64 w4 w3 w2
  ab 0 cfg5 1280 16 w4 w3 w2
  ab 0 materials 1024 16 w4 w3 w2
  ab 0 cornell1m 1024 16 w4 w3
  ab 0 plane 1280 64 w4 w3
  ab 2 corpus:features1 1280 64 w4 w3 w2
  ab 2 cfg5 1280 16 w4 w3 w2
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
