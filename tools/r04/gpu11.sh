#!/bin/bash
# configs[4] with SURVEY's camera: per-sample counts of the bench worker, and the cfg5 parity tests
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu11; mkdir -p $out
export TMPDIR=/tmp
{
  date
  timeout 150 python bench.py --worker configs4 --worker-json 2>&1 | grep YTHIP_RUN | cut -c1-900
  timeout 240 python -X faulthandler -m pytest tests/test_gpu_baseline_configs.py -x -q -k "cfg5" 2>&1 | grep -v "^  File \"/usr" | tail -8
  date
} > $out/hair.txt 2>&1
cat $out/hair.txt
