#!/bin/bash
# Round 6, GPU session 1: the suite on the tree as it came from round 5 (a fresh build), and the bound on what sorting rays
# between bounces can buy the traversal (tools/r06/sorted_extend_probe.py).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu1
out=gpurun_out/r06_gpu1; mkdir -p $out
export TMPDIR=/tmp
{
  date
  for sc in cfg2b configs4 configs3 features1 materials1; do
    RES=$([ $sc = configs3 ] && echo 1280 || echo "") 
    if [ -n "$RES" ]; then export RES; else unset RES; fi
    SCENE=$sc timeout 500 python tools/r06/sorted_extend_probe.py 2>&1 | grep -v "^\[bench"
  done
  date
} > $out/sorted_probe.txt 2>&1
cat $out/sorted_probe.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5 > $out/pytest.txt; cat $out/pytest.txt
