#!/bin/bash
# Round 5, GPU session 10: the full GPU suite on the candidate build, then bench.py as the driver runs it (line size, wall time).
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu10
out=gpurun_out/r05_gpu10; mkdir -p $out
export TMPDIR=/tmp
{ date; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15; date; } > $out/suite.txt 2>&1
{
  date; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
  t0=$(date +%s%N)
  python bench.py > $out/bench_line.json 2> $out/bench.err
  rc=$?
  t1=$(date +%s%N)
  echo "bench rc=$rc wall $(( (t1 - t0) / 1000000 )) ms, line bytes: $(wc -c < $out/bench_line.json)"
  cp bench_detail.json $out/bench_detail.json
  tail -5 $out/bench.err
} > $out/bench.txt 2>&1
cat $out/suite.txt $out/bench.txt; cat $out/bench_line.json
