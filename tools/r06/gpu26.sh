#!/bin/bash
# round 6, session 26: eviction of a wavefront's last walkers — parity, then A/B
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu26; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -n 12 > $out/pytest_stream.txt; cat $out/pytest_stream.txt
{
  date
  SCENES=configs4 VARIANTS=2:4:-1:2:250:0,2:4:-1:2:250:16,2:4:-1:2:250:24,2:4:-1:2:250:32,2:4:-1:2:250:24:16,2:4:-1:1:250:24 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=configs3 SPP=64 VARIANTS=1:3:-1:2:250:0,1:3:-1:2:250:16,1:3:-1:2:250:24,1:3:-1:2:250:32 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=features1,materials1 VARIANTS=2:4:-1:2:250:0,2:4:-1:2:250:16,2:4:-1:2:250:32 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=cfg2b VARIANTS=1:3:-1:2:250:0,1:3:0:2:250:0,1:3:0:2:250:16,1:3:0:2:250:32 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  date
} > $out/stream_ab_evict.txt 2>&1
cat $out/stream_ab_evict.txt
