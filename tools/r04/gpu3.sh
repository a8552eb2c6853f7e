#!/bin/bash
# Round 4, GPU session 3 (short): pretest / cooperative TLAS with the single-instance bypass, the opaque-textured class
# on the corpus scenes, the tolerance-mode tests.  Every piece under its own short timeout.  -> gpurun_out/r04_gpu3/
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu3; mkdir -p $out
export TMPDIR=/tmp
ab() {  # ab SCENE RES SPP lib...
  local sc=$1 res=$2 spp=$3; shift 3
  for n in "$@"; do
    printf "%-10s " $n
    YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=path DIGEST=1 timeout 90 python tools/sampler_times.py 2>&1 | grep -v "^\[timing\]" | tail -n 1
  done
}
{
  date
  ab plane 1280 64 base3 pre2 cooptlas2
  ab cornell1m 1024 16 base3 pre2 cooptlas2
  ab cfg4 1920 32 base3 pre2 cooptlas2
  ab cfg5 1280 16 base3 pre2 cooptlas2
  ab cornell9m 1024 16 base3 pre2 cooptlas2
  ab instances 1280 64 base3 pre2 cooptlas2
  ab lines_points 1280 64 base3 pre2 cooptlas2
  ab materials 1280 64 base3 pre2 cooptlas2
  ab cornellbox 1280 64 base3 pre2 cooptlas2
  echo "== the opaque-textured class (SPECIALIZE=0: the general class) on corpus scenes"
  for sc in corpus:materials1 corpus:materials3 corpus:features1 corpus:shapes1 corpus:arealights1; do
    for sp in 0 1; do printf "specialize=%d " $sp; SPECIALIZE=$sp YTHIP_LIB=$PWD/build/dev/libythip_base3.so SCENE=$sc RES=1280 SPP=64 SAMPLERS=path DIGEST=1 timeout 90 python tools/sampler_times.py 2>&1 | tail -n 1; done
    printf "pre2 spec=1  "; YTHIP_LIB=$PWD/build/dev/libythip_pre2.so SCENE=$sc RES=1280 SPP=64 SAMPLERS=path DIGEST=1 timeout 90 python tools/sampler_times.py 2>&1 | tail -n 1
  done
  date
} > $out/ab.txt 2>&1
{
  echo "== tolerance-mode tests"; date
  timeout 600 python -m pytest -x -q -s tests/test_gpu_fastmath.py 2>&1 | grep -v "^$" | tail -30
  date
} > $out/fast.txt 2>&1
cat $out/ab.txt $out/fast.txt
