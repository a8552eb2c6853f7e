#!/bin/bash
# Round 5, GPU session 25: the final tree once more — the whole GPU suite, smoke(), and bench.py exactly as the driver runs it.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu25
out=gpurun_out/r05_gpu25; mkdir -p $out
export TMPDIR=/tmp
{
  date
  timeout 1500 python -m pytest tests -m gpu -x -q > $out/suite_full.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $out/suite_full.log | tail -1
  date
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
  t0=$(date +%s%N)
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/line.json 2> $out/bench.err; echo "bench rc=$?"
  t1=$(date +%s%N)
  echo "bench wall $(( (t1 - t0) / 1000000 )) ms, line $(wc -c < $out/line.json) bytes"
  python -c "
import json
j=json.load(open('$out/line.json')); print('value', j['value'], 'ms', j['ms_per_step'], 'frac', j['roofline']['frac'], j['roofline']['fractions'], 'cpu', j['cpu_baseline']['value'])
for o in j['other_configs']: print(o)"
  date
} > $out/log.txt 2>&1
cat $out/log.txt
