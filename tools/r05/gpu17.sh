#!/bin/bash
# Round 5, GPU session 17: launch-overlap probe (tools/r05/overlap_probe.py): the frame as G chains of launches on G streams.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu17
out=gpurun_out/r05_gpu17; mkdir -p $out
export TMPDIR=/tmp
{
  date
  timeout 200 python tools/r05/overlap_probe.py plane 1280 64 20 2>&1 | grep -v "^\[" | tail -8
  FASTMATH=1 timeout 200 python tools/r05/overlap_probe.py plane 1280 64 20 2>&1 | grep -v "^\[" | tail -8
  timeout 300 python tools/r05/overlap_probe.py cornell1m 1024 16 10 2>&1 | grep -v "^\[" | tail -8
  timeout 300 python tools/r05/overlap_probe.py cfg5 1280 16 10 2>&1 | grep -v "^\[" | tail -8
  date
} > $out/probe.txt 2>&1
cat $out/probe.txt
