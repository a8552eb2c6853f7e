#!/bin/bash
# Round 6, GPU session 2: the streaming scheduler's first run — its parity tests, then fused vs streamed on the bench workloads.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu2
out=gpurun_out/r06_gpu2; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -n 30 > $out/pytest_stream.txt; cat $out/pytest_stream.txt
{
  date
  SCENES=cfg2b SPP=64 VARIANTS=0:4,1:4,2:4,0:3,0:5,0:4:0 timeout 600 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=configs4 SPP=64 VARIANTS=2:4,0:4,1:4 timeout 600 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=configs3 SPP=32 VARIANTS=2:4,0:4,1:4 timeout 600 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=features1,materials1 SPP=64 VARIANTS=2:4,0:4,1:4 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=configs1 SPP=64 VARIANTS=2:4,0:4 timeout 600 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  date
} > $out/stream_ab.txt 2>&1
cat $out/stream_ab.txt
