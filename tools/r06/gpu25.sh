#!/bin/bash
# round 6, session 25: what would evicting a wavefront's last walkers buy?  (per-ray walk lengths replayed on the host)
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu25; mkdir -p $out
export TMPDIR=/tmp
{
  SCENE=configs4 VARIANT=2:4 GENS=2,40,120 timeout 600 python tools/r06/regroup_sim.py
  SCENE=configs3 VARIANT=1:3 GENS=2,40,100 SPP=64 timeout 600 python tools/r06/regroup_sim.py
  SCENE=cfg2b VARIANT=1:3 GENS=2,40,120 timeout 600 python tools/r06/regroup_sim.py
  SCENE=features1 VARIANT=2:4 GENS=2,40,120 timeout 600 python tools/r06/regroup_sim.py
} > $out/regroup_sim.txt 2>&1
cat $out/regroup_sim.txt
