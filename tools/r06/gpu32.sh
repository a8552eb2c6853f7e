#!/bin/bash
# Round 6, GPU session 32: random-scene parity on the final kernels (the streamed cases draw the tail kernel's threshold and the walk form too)
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu32; mkdir -p $out
export TMPDIR=/tmp
for k in 0 1 2 3; do
  ( timeout 1500 python tools/fuzz_parity.py $((${FIRST:-40000} + k * ${COUNT:-1500})) ${COUNT:-1500} > $out/fuzz_$k.txt 2>&1 ) &
done
wait
{ date; for k in 0 1 2 3; do tail -n 1 $out/fuzz_$k.txt; done; grep -h "FAIL\|ERROR" $out/fuzz_*.txt | head -n 20; echo "streamed cases: $(grep -h 'streamed 1' $out/fuzz_*.txt | wc -l), of them through the tail kernel: $(grep -h 'streamed 1' $out/fuzz_*.txt | grep -v ': 0 rays' | wc -l)"; } | tee $out/fuzz_summary.txt
