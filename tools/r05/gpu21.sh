#!/bin/bash
# Round 5, GPU session 21: kernel trace of the chains probe: do the two chains' launches overlap in time?
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu21
out=$PWD/gpurun_out/r05_gpu21; mkdir -p $out
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt21 -- python $R/tools/r05/chains_probe.py plane 1280 64 6 > $out/probe.log 2>&1
f=$(find /tmp/kt21 -name '*kernel_trace.csv' | head -1)
python - "$f" > $out/trace.txt 2>&1 <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_trace' in r['Kernel_Name']]
print(len(rows), 'k_trace dispatches; columns', list(rows[0].keys())[:14])
t0 = int(rows[0]['Start_Timestamp'])
for r in rows[-40:]:
    print(r.get('Queue_Id'), r.get('Grid_Size', r.get('Grid_Size_X')), (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6)
PY
grep "chain(s)" $out/probe.log; cat $out/trace.txt | head -60
