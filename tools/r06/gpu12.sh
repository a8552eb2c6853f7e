#!/bin/bash
# Round 6, GPU session 12: the whole GPU suite on the product build (own-tree gates printed), then the round's evidence set.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu12
out=$PWD/gpurun_out/r06_gpu12; mkdir -p $out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q -s 2>&1 | grep "hits\]\|gate\]\|passed\|failed\|Error\|error\|assert" | sed 's/^\.*//' | cut -c1-400 > $out/pytest.txt; tail -n 60 $out/pytest.txt
bash tools/prof_r06.sh r06 > $out/prof.log 2>&1; tail -n 30 $out/prof.log
