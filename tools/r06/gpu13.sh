#!/bin/bash
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu13
out=$PWD/gpurun_out/r06_gpu13; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_own_tree.py -x -q -s -k "names_the_references_hits" 2>&1 | grep "hits\]\|passed\|failed\|Error\|assert" | sed 's/^\.*//' | cut -c1-500 > $out/own_hits.txt; cat $out/own_hits.txt
