#!/bin/bash
# Round 6, GPU session 38 (final build): the whole GPU suite, smoke, random-scene parity incl. naive / pathtest / pathdirect streamed
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu38
out=$PWD/gpurun_out/r06_gpu38; mkdir -p $out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -n 15 > $out/pytest.txt; cat $out/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
FIRST=60000 COUNT=2000 bash tools/r06/gpu32.sh
for s in naive pathtest pathdirect; do echo "$s streamed: $(grep -h " $s " gpurun_out/r06_gpu32/fuzz_*.txt | grep -c 'streamed 1')"; done | tee -a gpurun_out/r06_gpu32/fuzz_summary.txt
