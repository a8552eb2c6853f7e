#!/bin/bash
# Round 5, GPU session 19: ythip_set_batch_chains with the side stream at another priority — does a chain overlap the other now?
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu19
out=gpurun_out/r05_gpu19; mkdir -p $out
export TMPDIR=/tmp
{
  date
  for prio in greatest 0 -1 1; do
    for ts in 0 1; do
      if [ $prio = greatest ]; then unset YTHIP_CHAIN_PRIORITY; else export YTHIP_CHAIN_PRIORITY=$prio; fi
      TORCH_STREAM=$ts timeout 120 python tools/r05/chains_probe.py plane 1280 64 20 2>&1 | grep "chain(s)"
    done
  done
  unset YTHIP_CHAIN_PRIORITY
  timeout 200 python tools/r05/chains_probe.py cornell1m 1024 16 10 2>&1 | grep "chain(s)"
  python -c "
import ctypes
h=ctypes.CDLL('libamdhip64.so'); a=ctypes.c_int(); b=ctypes.c_int(); print('priority range rc', h.hipDeviceGetStreamPriorityRange(ctypes.byref(a), ctypes.byref(b)), 'least', a.value, 'greatest', b.value)"
  date
} > $out/log.txt 2>&1
cat $out/log.txt
