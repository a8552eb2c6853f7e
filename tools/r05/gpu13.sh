#!/bin/bash
# Round 5, GPU session 13: own-tree micro-variants against A (the shipped walk) — T branch-free triangle test (no early exits),
# L line / point tests on reciprocals + fused multiply-adds, P pushes without the LDS / spill decision when a ballot says all fit,
# TLP all three.  Then the hit agreement and the statistical gates of TLP.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu13
out=gpurun_out/r05_gpu13; mkdir -p $out
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
  ab cornell1m 1024 16 ownA ownT ownL ownP ownTLP ownA ownTLP
  ab cfg4 1920 16 ownA ownT ownP ownTLP
  ab cfg5 1280 16 ownA ownT ownL ownP ownTLP ownA ownTLP
  ab cornell9m 1024 16 ownA ownT ownP ownTLP
  ab corpus:materials1 1280 16 ownA ownT ownP ownTLP
  ab corpus:features1 1280 16 ownA ownT ownL ownP ownTLP
  ab lines_points 1024 16 ownA ownL ownTLP
  ab plane 1280 64 ownA ownT ownP ownTLP
  date
  YTHIP_LIB=$PWD/build/dev/libythip_ownTLP.so timeout 900 python -m pytest -q -s tests/test_gpu_own_tree.py 2>&1 | grep "hits\]\|gate\]\|passed\|failed" | sed 's/^\.*//'
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
