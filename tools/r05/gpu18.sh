#!/bin/bash
# Round 5, GPU session 18: ythip_set_batch_chains — its tests, the bench contract tests, the default bench line with the two-chains
# entry and timestamps of its progress, then the whole suite.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu18
out=gpurun_out/r05_gpu18; mkdir -p $out
export TMPDIR=/tmp
{
  date
  timeout 600 python -m pytest -q -x tests/test_gpu_round2.py -k "chains" 2>&1 | tail -5
  timeout 900 python -m pytest -q -x tests/test_gpu_bench_contract.py 2>&1 | tail -5
  date
  timeout 300 python bench.py --steps 20 --warmup 5 --detail $out/detail.json > $out/line.json 2> $out/bench.err; echo "bench rc=$?"
  grep "^\[bench" $out/bench.err | cut -c1-160
  python -c "
import json
j=json.load(open('$out/line.json')); print('value', j['value'], 'ms', j['ms_per_step'], 'two_chains', j.get('two_chains'), 'bytes', len(open('$out/line.json').read()))"
  date
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
  date
} > $out/log.txt 2>&1
cat $out/log.txt
