#!/bin/bash
# Round 5, GPU session 1: (a) first process under faultdump, (b) the A/B of the round-4 "kit" builds (whole-state digests
# must agree along a row), (c) the new bench line (size, wall time).  Output -> gpurun_out/r05_gpu1/
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu1
out=gpurun_out/r05_gpu1; mkdir -p $out
export TMPDIR=/tmp
LAUNCHES=6 bash tools/r05_first_session.sh > /dev/null 2>&1
cp gpurun_out/r05_first/ab.txt $out/ab.txt
{
  date; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -m1 "model name" /proc/cpuinfo
  /usr/bin/time -v python bench.py > $out/bench_line.json 2> $out/bench.err
  echo "bench rc=$? line bytes: $(wc -c < $out/bench_line.json)"; grep -E "Elapsed|Maximum resident" $out/bench.err
  cp bench_detail.json $out/bench_detail.json
  date
} > $out/bench.txt 2>&1
cat $out/ab.txt $out/bench.txt; head -c 4200 $out/bench_line.json
