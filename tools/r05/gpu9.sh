#!/bin/bash
# Round 5, GPU session 9: hit agreement of the variants with / without the slack factor (A: with; L: without).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu9
out=gpurun_out/r05_gpu9; mkdir -p $out
export TMPDIR=/tmp
for n in ownA ownL; do
  echo "== $n"
  YTHIP_LIB=$PWD/build/dev/libythip_$n.so timeout 600 python -m pytest -q -s tests/test_gpu_own_tree.py -k "names_the_references_hits" 2>&1 | grep "hits\]\|passed\|failed" | sed 's/^\.*//' | grep -v "per-instance"
done > $out/hits.txt 2>&1
cat $out/hits.txt
