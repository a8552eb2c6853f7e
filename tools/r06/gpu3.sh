#!/bin/bash
# Round 6, GPU session 3: streaming scheduler v2 (wave-aggregated histogram atomics, coalesced scan): tests, A/B, and the
# per-kernel split of a streamed batch (rocprofv3 --kernel-trace --stats).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu3
out=$PWD/gpurun_out/r06_gpu3; mkdir -p $out
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -n 30 > $out/pytest_stream.txt; cat $out/pytest_stream.txt
{
  date
  SCENES=cfg2b SPP=64 VARIANTS=0:4,1:4,2:4,0:3,1:3 timeout 600 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=configs4 SPP=64 VARIANTS=2:4,1:3 timeout 600 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=configs3 SPP=32 VARIANTS=2:4,1:3 timeout 600 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  SCENES=features1 SPP=64 VARIANTS=2:4,1:3 timeout 900 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  date
} > $out/stream_ab.txt 2>&1
cat $out/stream_ab.txt
cd /tmp
for sc in cfg2b configs4 features1; do
  v=$([ $sc = cfg2b ] && echo 1:3 || echo 2:4)
  SCENES=$sc SPP=64 VARIANTS=$v LAUNCHES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$sc -- python $R/tools/r06/stream_ab.py > $out/prof_$sc.log 2>&1
  find /tmp/kt_$sc -name '*_kernel_stats.csv' -exec cp {} $out/kernel_stats_$sc.csv \;
  head -n 12 $out/kernel_stats_$sc.csv | cut -c1-220
done
