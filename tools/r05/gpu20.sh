#!/bin/bash
# Round 5, GPU session 20: ythip_set_batch_chains with the chains put half a launch apart (the first batch after a join).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu20
out=gpurun_out/r05_gpu20; mkdir -p $out
export TMPDIR=/tmp
{
  date
  timeout 120 python tools/r05/chains_probe.py plane 1280 64 20 2>&1 | grep "chain(s)"
  timeout 120 python tools/r05/chains_probe.py plane 1280 64 64 2>&1 | grep "chain(s)"
  FASTMATH=1 timeout 120 python tools/r05/chains_probe.py plane 1280 64 20 2>&1 | grep "chain(s)"
  timeout 200 python tools/r05/chains_probe.py cornell1m 1024 16 12 2>&1 | grep "chain(s)"
  timeout 200 python tools/r05/chains_probe.py cfg5 1280 16 12 2>&1 | grep "chain(s)"
  timeout 200 python tools/r05/chains_probe.py cfg4 1920 16 12 2>&1 | grep "chain(s)"
  date
  timeout 600 python -m pytest -q -x tests/test_gpu_round2.py -k "chains" 2>&1 | grep -E "passed|failed"
  date
} > $out/log.txt 2>&1
cat $out/log.txt
