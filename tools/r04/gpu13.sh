#!/bin/bash
# Round 4, last session: the single-GPU probe of the N-GPU strong-scaling bound (tools/scale_probe_any.py)
# on the final build — rank 0's slice of N, no gather.
O=gpurun_out/r04_gpu13; mkdir -p $O
{
  SCENE=plane timeout 100 python tools/scale_probe_any.py
  SCENE=cfg4  timeout 200 python tools/scale_probe_any.py
} > $O/scale_probe.txt 2>&1
tail -8 $O/scale_probe.txt
