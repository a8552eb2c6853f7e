#!/bin/bash
# Round 6, GPU session 36 (final build): the whole GPU suite, smoke, the evidence set, the measured choice per workload (path, pathdirect)
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu36
out=$PWD/gpurun_out/r06_gpu36; mkdir -p $out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -n 15 > $out/pytest.txt; cat $out/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
bash tools/prof_r06.sh r06 2>&1 | tail -n 8
{ date; timeout 1200 python tools/r06/choice_ab.py; FASTMATH=2 SCENES=cfg2b,configs3,configs4,cornell9m timeout 600 python tools/r06/choice_ab.py; SAMPLER=pathdirect SCENES=cfg2b,configs3,cornell9m,materials1 SPP=32 timeout 900 python tools/r06/choice_ab.py; date; } > $out/choice_ab.txt 2>&1
cat $out/choice_ab.txt
