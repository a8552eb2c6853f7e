#!/bin/bash
# Round 5, GPU session 24: the pixel pool's workgroup count (YTHIP_POOL_BLOCKS; default 16 per CU = 4,096 = the resident slots).
# Session 17's two half-frame contexts ran 2 x 4,096 pool workgroups for one frame and were 4 % faster: more workgroups than slots?
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu24
out=gpurun_out/r05_gpu24; mkdir -p $out
export TMPDIR=/tmp
run() {
  local sc=$1 res=$2 spp=$3; shift 3
  for pb in "$@"; do
    printf "pool_blocks %-6s " $pb
    DIGEST=1 YTHIP_PIXEL_POOL=2 YTHIP_POOL_BLOCKS=$pb SCENE=$sc RES=$res SPP=$spp SAMPLERS=path LAUNCHES=6 timeout 120 python tools/sampler_times.py 2>&1 | grep -v "^\[" | tail -n 1
  done
  printf "pool off           "
  DIGEST=1 YTHIP_PIXEL_POOL=0 SCENE=$sc RES=$res SPP=$spp SAMPLERS=path LAUNCHES=6 timeout 120 python tools/sampler_times.py 2>&1 | grep -v "^\[" | tail -n 1
}
{
  date
  run plane 1280 64 2048 3072 4096 5120 6144 8192 12288
  run cornell1m 1024 64 3072 4096 6144 8192 12288
  run cfg5 1280 64 3072 4096 6144 8192
  run cfg4 1920 64 3072 4096 6144 8192 16384
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
