#!/bin/bash
# Round 6, GPU session 8: the three parity-level walk leads — seven-load quad records (YT_WIDE7), the per-shape
# half of the instance records as launch constants (YT_ONE_SHAPE), shading positions from the hit's leaf record (YT_LEAF_SHADE):
# each switched off alone against all three on, whole-state digests; then the whole GPU suite on the product build.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu8
out=$PWD/gpurun_out/r06_gpu8; mkdir -p $out
export TMPDIR=/tmp
{
  date
  SCENES="plane cornell1m cfg4 cfg5 corpus:features1 corpus:materials1 cornell9m" bash tools/ab_libs.sh r6alloff r6all r6w7off r6oneoff r6leafoff r6alloff r6all
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 15 > $out/pytest.txt; cat $out/pytest.txt
