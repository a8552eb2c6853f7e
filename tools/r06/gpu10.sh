#!/bin/bash
# Round 6, GPU session 10: the fed extend stage (persistent wavefronts that refill idle lanes from the sorted queue).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu10
out=$PWD/gpurun_out/r06_gpu10; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -n 30 > $out/pytest_stream.txt; cat $out/pytest_stream.txt
{
  date
  SCENES="plane cornell1m cfg4 cfg5" bash tools/ab_libs.sh r6b r6c r6b r6c
  SCENES=cfg2b SPP=64 LAUNCHES=3 VARIANTS=1:3:-1:2:0,1:3:-1:2:4096,1:3:-1:2:5120,1:3:-1:2:8192,1:3:-1:1:4096,1:3:0:2:4096,2:4:-1:2:4096 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=configs4 SPP=64 LAUNCHES=2 VARIANTS=2:4:-1:2:0,2:4:-1:2:4096,2:4:-1:2:8192,2:4:-1:1:4096,1:3:-1:2:4096 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=configs3 SPP=64 LAUNCHES=2 VARIANTS=1:3:-1:2:0,1:3:-1:2:4096,1:3:-1:2:8192,1:3:-1:1:4096 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=features1,materials1,cornell9m SPP=64 LAUNCHES=2 VARIANTS=2:4:-1:2:0,2:4:-1:2:4096,1:3:-1:2:4096 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  date
} > $out/stream_ab.txt 2>&1
cat $out/stream_ab.txt
