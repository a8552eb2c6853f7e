#!/bin/bash
# round 6, session 23: ks_scan as four wavefronts — stream tests + A/B on the stream workloads
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu23; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -n 5 > $out/pytest_stream.txt; cat $out/pytest_stream.txt
{
  date
  SCENES=cfg2b VARIANTS=1:3:-1:2:0,1:3:-1:2:250,1:3:-1:3:250,1:3:-1:1:250,1:4:-1:2:250,0:4:-1:2:250 LAUNCHES=3 timeout 600 python tools/r06/stream_ab.py
  SCENES=configs3 VARIANTS=1:3:-1:2:0,1:3:-1:2:250,1:3:-1:3:250 LAUNCHES=3 timeout 600 python tools/r06/stream_ab.py
  SCENES=configs4 VARIANTS=2:4:-1:2:0,2:4:-1:2:250,2:4:-1:3:250 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=features1,materials1 VARIANTS=2:4:-1:2:250,2:4:-1:2:1000 LAUNCHES=2 timeout 600 python tools/r06/stream_ab.py
  SCENES=cornell9m VARIANTS=1:3:-1:2:0,1:3:-1:2:250 LAUNCHES=3 timeout 600 python tools/r06/stream_ab.py
  date
} > $out/stream_ab_scan.txt 2>&1
cat $out/stream_ab_scan.txt
SCENES=cfg2b FINS=250 bash tools/r06/gpu22.sh
