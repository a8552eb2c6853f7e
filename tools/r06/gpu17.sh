#!/bin/bash
# round 6, session 3: counters of the own-tree kernels, fused and streamed (what binds the own walk when it is alone in a kernel?)
out=gpurun_out/r06_gpu17; mkdir -p $out
export TMPDIR=/tmp
{
  FASTMATH=2 PMC_TIMEOUT=300 timeout 900 python tools/r06/stream_pmc.py cfg2b 16
  FASTMATH=2 VARIANT=2:4 PMC_TIMEOUT=300 timeout 900 python tools/r06/stream_pmc.py configs4 16
  FASTMATH=2 PMC_TIMEOUT=300 timeout 900 python tools/r06/stream_pmc.py configs3 16
} > $out/stream_pmc_own.txt 2>&1
cat $out/stream_pmc_own.txt
