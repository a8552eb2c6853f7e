#!/bin/bash
# round 6, session 3: the parity fuzz at scale on the final kernels (seven-load records, leaf-record shading, NEE classes, streaming scheduler)
out=gpurun_out/r06_gpu18; mkdir -p $out
{ date; timeout 1500 python tools/fuzz_parity.py 20000 12000; date; } > $out/fuzz_full.txt 2>&1
{ head -4 $out/fuzz_full.txt; grep -c "streamed 1" $out/fuzz_full.txt | sed 's/^/cases that ran on the streaming scheduler: /'; grep -v ": ok " $out/fuzz_full.txt | tail -40; } > $out/fuzz.txt
cat $out/fuzz.txt
