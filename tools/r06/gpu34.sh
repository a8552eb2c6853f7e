#!/bin/bash
# round 6, session 34: pathdirect on the streaming scheduler (NEE half inline in the shade stage) — parity, then A/B
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu34; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -n 12 > $out/pytest_stream.txt; cat $out/pytest_stream.txt
{
  date
  SAMPLER=pathdirect SCENES=cfg2b SPP=16 VARIANTS=0:4,0:4:-1:2:0,2:4 LAUNCHES=3 timeout 600 python tools/r06/stream_ab.py
  SAMPLER=pathdirect SCENES=cornell9m,configs4,configs3,materials1 SPP=16 VARIANTS=0:4 LAUNCHES=2 timeout 900 python tools/r06/stream_ab.py
  date
} > $out/stream_ab_pathdirect.txt 2>&1
cat $out/stream_ab_pathdirect.txt
