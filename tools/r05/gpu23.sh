#!/bin/bash
# Round 5, GPU session 23: the workgroup size again, with today's walks (round 1 measured 256 / 128 / 64 threads on the first walk and
# chose 64): -DYT_BLOCK_SIZE=128 / 256 — the block-local queue then packs camera rays and bounce rays of 2 / 4 wavefronts' slots
# into wavefronts of one kind.  Bit-exact (digests) and on the own tree.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu23
out=gpurun_out/r05_gpu23; mkdir -p $out
export TMPDIR=/tmp
ab() {
  local fm=$1 sc=$2 res=$3 spp=$4; shift 4
  for n in "$@"; do
    printf "%-8s fastmath %s " $n $fm
    DIGEST=1 FASTMATH=$fm YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=path LAUNCHES=5 timeout 120 python tools/sampler_times.py 2>&1 | grep -v "^\[" | tail -n 1
  done
}
{
  date
  for fm in 0 2; do
    ab $fm cornell1m 1024 16 blk64 blk128 blk256
    ab $fm cfg4 1920 16 blk64 blk128 blk256
    ab $fm cfg5 1280 16 blk64 blk128 blk256
    ab $fm corpus:features1 1280 16 blk64 blk128 blk256
    ab $fm plane 1280 64 blk64 blk128 blk256
  done
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
