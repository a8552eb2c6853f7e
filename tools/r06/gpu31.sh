#!/bin/bash
# Round 6, GPU session 31: the whole GPU suite with the measured choice of scheduler as the library's default, then the evidence set
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu31
out=$PWD/gpurun_out/r06_gpu31; mkdir -p $out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -n 15 > $out/pytest.txt; cat $out/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3
bash tools/prof_r06.sh r06 2>&1 | tail -n 30
