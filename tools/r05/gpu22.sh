#!/bin/bash
# Round 5, GPU session 22: the chains probe with the pixel pool off (a pool launch goes out whole: session 21 showed the pool on).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu22
out=gpurun_out/r05_gpu22; mkdir -p $out
export TMPDIR=/tmp
{
  date
  for pool in 0 1; do
    echo "== YTHIP_PIXEL_POOL=$pool"
    YTHIP_PIXEL_POOL=$pool timeout 120 python tools/r05/chains_probe.py plane 1280 64 20 2>&1 | grep "chain(s)"
    YTHIP_PIXEL_POOL=$pool FASTMATH=1 timeout 120 python tools/r05/chains_probe.py plane 1280 64 20 2>&1 | grep "chain(s)"
    YTHIP_PIXEL_POOL=$pool timeout 200 python tools/r05/chains_probe.py cornell1m 1024 16 12 2>&1 | grep "chain(s)"
    YTHIP_PIXEL_POOL=$pool timeout 200 python tools/r05/chains_probe.py cfg5 1280 16 12 2>&1 | grep "chain(s)"
    YTHIP_PIXEL_POOL=$pool timeout 200 python tools/r05/chains_probe.py cfg4 1920 16 12 2>&1 | grep "chain(s)"
  done
  date
} > $out/log.txt 2>&1
cat $out/log.txt
