#!/bin/bash
# round 6, session 28: eviction removed, majority-phase walk off by default in ks_extend — stream tests + library defaults on the workloads
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu28; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -n 12 > $out/pytest_stream.txt; cat $out/pytest_stream.txt
{
  date
  SCENES=cfg2b VARIANTS=0:4,1:3,0:3,0:5 LAUNCHES=3 timeout 600 python tools/r06/stream_ab.py
  SCENES=configs4 VARIANTS=0:4,2:4,0:3 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=configs3 SPP=64 VARIANTS=0:4,1:3,2:4 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=cornell9m VARIANTS=0:4,1:3 LAUNCHES=3 timeout 600 python tools/r06/stream_ab.py
  FASTMATH=1 SCENES=cfg2b,configs4,cornell9m VARIANTS=0:4 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  date
} > $out/stream_ab_defaults.txt 2>&1
cat $out/stream_ab_defaults.txt
