#!/bin/bash
# Round 5, GPU session 26: the general kernel class (CLS 0, TRI 0) without the scalar-operand copy of the wide step
# (-DYT_GENERAL_VECTOR_WALK): fewer registers there?  Bit-exact: digests.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu26
out=gpurun_out/r05_gpu26; mkdir -p $out
export TMPDIR=/tmp
ab() {
  local sc=$1 res=$2 spp=$3; shift 3
  for n in "$@"; do
    printf "%-8s " $n
    DIGEST=1 YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=path LAUNCHES=5 timeout 120 python tools/sampler_times.py 2>&1 | grep -v "^\[" | tail -n 1
  done
}
{
  date
  ab corpus:features1 1280 64 gbase gvec gbase gvec
  ab corpus:shapes1 1280 64 gbase gvec
  ab materials 1024 16 gbase gvec
  ab lines_points 1024 16 gbase gvec
  ab cfg5 1280 16 gbase gvec gbase gvec
  ab corpus:materials1 1280 64 gbase gvec
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
