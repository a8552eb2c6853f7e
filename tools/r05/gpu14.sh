#!/bin/bash
# Round 5, GPU session 14: the TLAS-leaf pretest on WORLD boxes of the instance roots (W: two loads + one slab test per instance,
# the transform only for whoever passes) against L2 (= the shipped walk + own_line).  Then hit agreement and gates of W.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu14
out=gpurun_out/r05_gpu14; mkdir -p $out
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
  ab cfg4 1920 16 ownL2 ownW ownL2 ownW
  ab corpus:features1 1280 16 ownL2 ownW ownL2 ownW
  ab corpus:materials1 1280 16 ownL2 ownW
  ab corpus:shapes1 1280 16 ownL2 ownW
  ab instances 1024 16 ownL2 ownW
  ab materials 1024 16 ownL2 ownW
  ab cornell1m 1024 16 ownL2 ownW
  ab cfg5 1280 16 ownL2 ownW
  ab plane 1280 64 ownL2 ownW
  date
  YTHIP_LIB=$PWD/build/dev/libythip_ownW.so timeout 900 python -m pytest -q -s tests/test_gpu_own_tree.py 2>&1 | grep "hits\]\|gate\]\|passed\|failed" | sed 's/^\.*//'
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
