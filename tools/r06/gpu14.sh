#!/bin/bash
# Round 6, GPU session 14: the whole GPU suite (gates printed).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu14
out=$PWD/gpurun_out/r06_gpu14; mkdir -p $out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q -s 2>&1 | grep "hits\]\|gate\]\|passed\|failed\|Error\|error\|assert" | sed 's/^\.*//' | cut -c1-600 > $out/pytest.txt; tail -n 40 $out/pytest.txt
