#!/bin/bash
# Round 6, GPU session 7: streaming scheduler micro-variants (development builds): non-temporal state I/O, 5 waves per SIMD,
# more hardware queues for 3-4 groups.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu7
out=$PWD/gpurun_out/r06_gpu7; mkdir -p $out
export TMPDIR=/tmp
{
  date
  for lib in r6base r6nt r6w5; do
    echo "--- $lib"
    for sc in cfg2b configs4 configs3; do
      v=$([ $sc = configs4 ] && echo 2:4:-1:2:1 || echo 1:3:-1:2:1)
      YTHIP_LIB=$PWD/build/dev/libythip_$lib.so SCENES=$sc SPP=64 LAUNCHES=3 VARIANTS=$v timeout 600 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
    done
  done
  echo "--- GPU_MAX_HW_QUEUES=8"
  GPU_MAX_HW_QUEUES=8 YTHIP_LIB=$PWD/build/dev/libythip_r6base.so SCENES=cfg2b SPP=64 LAUNCHES=3 VARIANTS=1:3:-1:2:1,1:3:-1:3:1,1:3:-1:4:1,1:3:-1:6:1 timeout 600 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  GPU_MAX_HW_QUEUES=8 YTHIP_LIB=$PWD/build/dev/libythip_r6base.so SCENES=configs4 SPP=64 LAUNCHES=2 VARIANTS=2:4:-1:2:1,2:4:-1:4:1 timeout 600 python tools/r06/stream_ab.py 2>&1 | grep -v "^\[bench"
  date
} > $out/variants.txt 2>&1
cat $out/variants.txt
