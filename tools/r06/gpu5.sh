#!/bin/bash
# Round 6, GPU session 5: streaming scheduler v3 — two groups on two streams, the pixel queue (fewer path slots than pixels).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu5
out=$PWD/gpurun_out/r06_gpu5; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -n 30 > $out/pytest_stream.txt; cat $out/pytest_stream.txt
{
  date
  SCENES=cfg2b SPP=64 LAUNCHES=3 VARIANTS=1:3,1:3:-1:2:1,1:3:-1:1:0.5,1:3:-1:2:0.5,1:3:-1:1:0.375,1:3:-1:2:0.375,1:3:-1:2:0.25 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=configs4 SPP=64 LAUNCHES=2 VARIANTS=2:4,2:4:-1:2:1,2:4:-1:1:0.5,2:4:-1:2:0.5,2:4:-1:2:0.3 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=configs3 SPP=64 LAUNCHES=2 VARIANTS=1:3,1:3:-1:2:1,1:3:-1:2:0.5,1:3:-1:2:0.25 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=features1,materials1 SPP=64 LAUNCHES=2 VARIANTS=2:4,2:4:-1:2:1,2:4:-1:2:0.4 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=cornell9m LAUNCHES=2 VARIANTS=1:3,1:3:-1:2:1,1:3:-1:2:0.5 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  date
} > $out/stream_ab.txt 2>&1
cat $out/stream_ab.txt
