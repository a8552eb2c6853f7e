#!/bin/bash
# Round 6, GPU session 6: streaming scheduler — more groups; counters of its kernels next to the fused kernel's.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu6
out=$PWD/gpurun_out/r06_gpu6; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -n 30 > $out/pytest_stream.txt; cat $out/pytest_stream.txt
{
  date
  SCENES=cfg2b SPP=64 LAUNCHES=3 VARIANTS=1:3:-1:1:1,1:3:-1:2:1,1:3:-1:3:1,1:3:-1:4:1,1:3:-1:6:1,1:3:-1:8:1,0:4:-1:4:1,1:3:0:4:1 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=configs4 SPP=64 LAUNCHES=2 VARIANTS=2:4:-1:2:1,2:4:-1:3:1,2:4:-1:4:1,2:4:-1:6:1,2:4:-1:8:1,1:3:-1:4:1 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=configs3 SPP=64 LAUNCHES=2 VARIANTS=1:3:-1:2:1,1:3:-1:4:1,1:3:-1:8:1,2:4:-1:4:1 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=features1,materials1 SPP=64 LAUNCHES=2 VARIANTS=2:4:-1:2:1,2:4:-1:4:1,2:4:-1:8:1 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=cornell9m,configs1 LAUNCHES=2 VARIANTS=1:3:-1:2:1,1:3:-1:4:1,1:3:-1:8:1 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  date
} > $out/stream_ab.txt 2>&1
cat $out/stream_ab.txt
{
  PMC_TIMEOUT=300 timeout 1300 python tools/r06/stream_pmc.py cfg2b 16
  VARIANT=2:4 PMC_TIMEOUT=300 timeout 1300 python tools/r06/stream_pmc.py configs4 16
  VARIANT=2:4 PMC_TIMEOUT=300 timeout 1300 python tools/r06/stream_pmc.py features1 16
} > $out/stream_pmc.txt 2>&1
cat $out/stream_pmc.txt
