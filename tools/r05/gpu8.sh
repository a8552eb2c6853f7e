#!/bin/bash
# Round 5, GPU session 8: own-tree variants — S speculative leaf postponement, L lean node step (scales as floats, no slack
# factor), 5 = 5 waves per SIMD with 4 LDS stack levels, SL both; against A (the current default).  + hit agreement of L.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu8
out=gpurun_out/r05_gpu8; mkdir -p $out
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
  ab cornell1m 1024 16 ownA ownS ownL own5 ownSL ownA
  ab cfg4 1920 16 ownA ownS ownL own5 ownSL
  ab cfg5 1280 16 ownA ownS ownL own5 ownSL
  ab cornell9m 1024 16 ownA ownS ownL own5
  ab corpus:materials1 1280 16 ownA ownS ownL own5 ownSL
  ab corpus:features1 1280 16 ownA ownS ownL own5 ownSL
  ab plane 1280 64 ownA ownS ownL own5 ownSL
  date
  YTHIP_LIB=$PWD/build/dev/libythip_ownSL.so timeout 600 python -m pytest -q -s tests/test_gpu_own_tree.py -k "names_the_references_hits" 2>&1 | grep "hits\]\|passed\|failed" | sed 's/^\.*//'
} > $out/ab.txt 2>&1
cat $out/ab.txt
