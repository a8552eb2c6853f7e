#!/bin/bash
# The first GPU session of round 5, prepared at the end of round 4 ("reading the ISA").  Build on the CPU box first
# (each step adds to the one before; about 1.5 minutes of hipcc each):
#   tools/devbuild.sh base
#   tools/devbuild.sh whole   -DYT_WHOLE_RECORD=1
#   tools/devbuild.sh byvalue -DYT_WHOLE_RECORD=1 -DYT_RECORDS_BY_VALUE
#   tools/devbuild.sh libm    -DYT_WHOLE_RECORD=1 -DYT_RECORDS_BY_VALUE -DYT_LIBM_NO_TABLES
#   tools/devbuild.sh tex     -DYT_WHOLE_RECORD=1 -DYT_RECORDS_BY_VALUE -DYT_LIBM_NO_TABLES -DYT_SRGB_LUT -DYT_TEXELS_TOGETHER
#   tools/devbuild.sh surface -DYT_WHOLE_RECORD=1 -DYT_RECORDS_BY_VALUE -DYT_LIBM_NO_TABLES -DYT_SRGB_LUT -DYT_TEXELS_TOGETHER -DYT_SURFACE_BY_VALUE
# then  gpurun --timeout 300 -- 'bash tools/r05_first_session.sh'  (about a minute of GPU).  Whole-state digests must agree along
# every row (the builds differ in WHEN things are fetched, never in what is computed); adopt what wins by passing the macro(s) in
# __graft_entry__'s flags (or dropping the #ifdefs) and run the full GPU suite before committing.
cd "$(dirname "$0")/.."
out=gpurun_out/r05_first; mkdir -p $out
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
  ab plane 1280 64 base whole byvalue libm base whole byvalue libm
  ab cornell1m 1024 16 base whole byvalue libm base whole byvalue libm
  ab cfg4 1920 16 base whole byvalue libm
  ab cfg5 1280 16 base whole byvalue libm
  ab cornell9m 1024 16 base whole byvalue
  ab corpus:materials1 1280 16 base byvalue libm tex surface base byvalue libm tex surface
  ab corpus:features1 1280 16 base byvalue libm tex surface base byvalue libm tex surface
  ab corpus:shapes1 1280 16 base byvalue tex surface
  ab materials 1280 16 base byvalue tex surface
  SAMPLERS=pathdirect ab cornell1m 1024 16 base byvalue
  SAMPLERS=pathmis ab corpus:materials1 1280 16 base tex
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
