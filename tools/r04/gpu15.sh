#!/bin/bash
# Round 4, GPU session 15 (the round's last GPU minute): -DYT_WHOLE_RECORD — the wide step's record kept as eight whole float4
# loads.  hipcc narrows the loads to the words used and SINKS slot a's ref (offset 24) into the branch behind the slab tests:
# a ninth, dependent load and an s_waitcnt vmcnt(0) in every step whose first slot passes.
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu15; mkdir -p $out
export TMPDIR=/tmp LAUNCHES=4
ab() {
  local sc=$1 res=$2 spp=$3; shift 3
  for n in "$@"; do
    printf "%-10s " $n
    YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=path DIGEST=1 timeout 20 python tools/sampler_times.py 2>&1 | grep -v "^\[timing\]" | tail -n 1
  done
}
{
  ab plane 1280 64 base9 whole base9 whole
  ab cfg4 1920 16 base9 whole
  ab cornell1m 1024 16 base9 whole
  ab cfg5 1280 16 base9 whole
} > $out/ab.txt 2>&1
cat $out/ab.txt
