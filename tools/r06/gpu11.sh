#!/bin/bash
# Round 6, GPU session 11: the fed extend stage by refill threshold (8 / 24 / 48 / 64 idle lanes; 64 = a persistent wavefront
# that takes its next 64 rays only when all are done, i.e. no refill at all) — where does its factor of two go?
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu11
out=$PWD/gpurun_out/r06_gpu11; mkdir -p $out
export TMPDIR=/tmp
{
  date
  for lib in r6c r6f8 r6f48 r6f64; do
    echo "--- $lib (threshold: r6c 24)"
    YTHIP_LIB=$PWD/build/dev/libythip_$lib.so SCENES=cfg2b SPP=64 LAUNCHES=2 VARIANTS=1:3:-1:2:0,1:3:-1:2:4096 timeout 600 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench" | grep stream
    YTHIP_LIB=$PWD/build/dev/libythip_$lib.so SCENES=configs4 SPP=64 LAUNCHES=2 VARIANTS=2:4:-1:2:0,2:4:-1:2:4096 timeout 600 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench" | grep stream
  done
  date
} > $out/feed_thresh.txt 2>&1
cat $out/feed_thresh.txt
