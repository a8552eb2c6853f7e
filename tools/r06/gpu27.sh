#!/bin/bash
# round 6, session 27: eviction with the wave-uniform condition; the majority-phase walk off in ks_extend
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu27; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -n 12 > $out/pytest_stream.txt; cat $out/pytest_stream.txt
{
  date
  SCENES=configs4 VARIANTS=2:4:-1:2:250:0,2:4:-1:2:250:8:3,2:4:-1:2:250:16:3,2:4:-1:2:250:24:3,2:4:-1:2:250:16:6 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=configs3 SPP=64 VARIANTS=1:3:-1:2:250:0,1:3:-1:2:250:8:3,1:3:-1:2:250:16:3,1:3:-1:2:250:24:3 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=features1 VARIANTS=2:4:-1:2:250:0,2:4:-1:2:250:16:3 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=cfg2b VARIANTS=1:3:-1:2:250:0,1:3:0:2:250:0,1:3:0:3:250:0,0:4:0:2:250:0,1:4:0:2:250:0,1:3:0:2:250:16:3 LAUNCHES=3 timeout 600 python tools/r06/stream_ab.py
  SCENES=cornell9m VARIANTS=1:3:-1:2:250:0,1:3:0:2:250:0,1:3:1:2:250:0 LAUNCHES=3 timeout 600 python tools/r06/stream_ab.py
  FASTMATH=2 SCENES=cfg2b,cornell9m VARIANTS=1:3:-1:2:250:0,1:3:0:2:250:0,1:3:1:2:250:0 LAUNCHES=3 timeout 600 python tools/r06/stream_ab.py
  date
} > $out/stream_ab_evict.txt 2>&1
cat $out/stream_ab_evict.txt
