#!/bin/bash
# round 6, session 39: how the groups' tiles are dealt (chunks of 2^k tiles round-robin; 30 = contiguous runs)
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu39; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -q 2>&1 | tail -n 2
{
  date
  for c in 30 12 10 8 6 4 0; do
    echo "--- YTHIP_STREAM_CHUNK=$c"
    YTHIP_STREAM_CHUNK=$c SCENES=cfg2b,configs4,cornell9m,features1,materials1,configs1 VARIANTS=1:3 LAUNCHES=3 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep "stream order"
    YTHIP_STREAM_CHUNK=$c SCENES=configs3 SPP=64 VARIANTS=1:3 LAUNCHES=2 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep "stream order"
  done
  date
} > $out/stream_ab_chunks.txt 2>&1
cut -c1-140 $out/stream_ab_chunks.txt
