#!/bin/bash
# Round 5, GPU session 28: the own tree with larger leaves (YTHIP_OWN_MAX_LEAF = 4 ... 7 primitives; the reference's trees stop at 4).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu28
out=gpurun_out/r05_gpu28; mkdir -p $out
export TMPDIR=/tmp
run() {
  local sc=$1 res=$2 spp=$3; shift 3
  for ml in "$@"; do
    printf "max leaf %s  " $ml
    YTHIP_OWN_MAX_LEAF=$ml FASTMATH=2 YTHIP_LIB=$PWD/build/dev/libythip_leaf.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=path LAUNCHES=5 timeout 120 python tools/sampler_times.py 2>&1 | grep -v "^\[bench" | tail -n 2 | tr '\n' ' '; echo
  done
}
{
  date
  run cornell1m 1024 16 4 5 6 7 4
  run cfg4 1920 16 4 5 6 7
  run cfg5 1280 16 4 5 6 7
  run cornell9m 1024 16 4 6 7
  run corpus:features1 1280 16 4 5 6 7
  run corpus:materials1 1280 16 4 6 7
  run plane 1280 64 4 6 7
  date
  YTHIP_OWN_MAX_LEAF=7 YTHIP_LIB=$PWD/build/dev/libythip_leaf.so timeout 600 python -m pytest -q -s tests/test_gpu_own_tree.py -k "names_the_references_hits" 2>&1 | grep "hits\]\|passed\|failed" | sed 's/^\.*//' | cut -c1-200
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
