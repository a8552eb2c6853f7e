#!/bin/bash
# round 6, session 24: the measured choice of scheduler (ythip_set_scheduler 2) — stream tests, then what it decides per workload
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu24; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -n 12 > $out/pytest_stream.txt; cat $out/pytest_stream.txt
{ date; timeout 1200 python tools/r06/choice_ab.py; FASTMATH=2 SCENES=cfg2b,configs3,configs4,cornell9m timeout 600 python tools/r06/choice_ab.py; date; } > $out/choice_ab.txt 2>&1
cat $out/choice_ab.txt
