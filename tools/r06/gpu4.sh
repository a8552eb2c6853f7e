#!/bin/bash
# Round 6, GPU session 4: the shape of a streamed batch (queue length and kernel times per generation), stream tests.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu4
out=$PWD/gpurun_out/r06_gpu4; mkdir -p $out
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -n 30 > $out/pytest_stream.txt; cat $out/pytest_stream.txt
cd /tmp
for sc in cfg2b configs4 configs3; do
  v=$([ $sc = cfg2b ] && echo 1:3 || echo 2:4)
  spp=$([ $sc = configs3 ] && echo 64 || echo 64)
  rm -rf /tmp/kt_$sc
  SCENE=$sc SPP=$spp VARIANT=$v GEN_OUT=/tmp/gen_$sc.npy timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$sc -- python $R/tools/r06/gen_profile.py 2>&1 | grep -v "^[WEI]2026\|^\[bench" > $out/gen_$sc.txt
  python $R/tools/r06/gen_join.py /tmp/kt_$sc /tmp/gen_$sc.npy >> $out/gen_$sc.txt 2>&1
  cat $out/gen_$sc.txt
done
