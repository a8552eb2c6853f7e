#!/bin/bash
# Round 6, GPU session 30: the whole GPU suite on the build with the tail kernel / measured choice; then once more with the measured
# choice as every context's default (YTHIP_SCHEDULER=2: a robustness run — tests that assert which scheduler ran may fail, parity must not)
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu30
out=$PWD/gpurun_out/r06_gpu30; mkdir -p $out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 15 > $out/pytest.txt; cat $out/pytest.txt
YTHIP_SCHEDULER=2 timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -n 40 > $out/pytest_choice_default.txt; cat $out/pytest_choice_default.txt
