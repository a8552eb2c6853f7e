#!/bin/bash
# Round-6 evidence for bench.py's line — ONE session, one process order, so that the files reproduce each other:
#   1. clocks / power of the idle device (rocm-smi), then
#   2. `python bench.py --steps 20 --warmup 5` UNTRACED (the line + its side file with the raw counters of its own --pmc passes),
#   3. `rocprofv3 --kernel-trace --stats` of the primary workload alone (the dominant kernel's average duration) and of the full
#      line (every kernel, incl. the streamed ks_* kernels and the NEE class kernels),
#   4. clocks / power again, right after the last launch,
#   5. tools/check_bench_line.py on the three files: fails if the line's fractions, traffic or launch time differ from what the
#      counters and the trace say by more than 2 % (the trace: 4 %, tracing costs a 5-ms launch that much),
#   6. the line as the driver prints it (default flags), kernel resources, the streamed kernels' counters.
# Run on the GPU box from the repo root:  bash tools/prof_r06.sh [tag]  ->  gpurun_out/profiles_<tag>/ (copy into profiles/)
TAG=${1:-r06}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/profiles_$TAG
rm -rf $O; mkdir -p $O
smi() { { date; rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -v "^=\|^$" | head -n 40; } > $O/${TAG}_smi_$1.txt; }
smi before
timeout 900 python bench.py --steps 20 --warmup 5 --detail $O/${TAG}_bench_detail.json > $O/${TAG}_bench.log 2>&1
grep '^{' $O/${TAG}_bench.log > $O/${TAG}_bench.json
smi after_bench
cd /tmp
export YTHIP_LPT_PROBE=0  # (the traces: no 1 + 63-sample split of a first batch in the kernel's average)
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- \
  python $R/bench.py --steps 20 --warmup 5 --no-counters --no-cpu-baseline --no-other-configs --detail /tmp/detail_a.json > $O/${TAG}_bench_under_rocprof.log 2>&1
find /tmp/kt_$TAG -name '*_kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats.csv \;
grep '^{' $O/${TAG}_bench_under_rocprof.log > $O/${TAG}_bench_under_rocprof.json
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kta_$TAG -- \
  python $R/bench.py --steps 3 --warmup 1 --no-counters --no-cpu-baseline --detail /tmp/detail_b.json > /dev/null 2>&1
find /tmp/kta_$TAG -name '*_kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats_all.csv \;
unset YTHIP_LPT_PROBE
cd $R
smi after_traces
python tools/check_bench_line.py $O/${TAG}_bench.json $O/${TAG}_bench_detail.json $O/${TAG}_kernel_stats.csv > $O/${TAG}_check_bench_line.txt 2>&1
echo "check_bench_line rc=$?" >> $O/${TAG}_check_bench_line.txt
cat $O/${TAG}_check_bench_line.txt
t0=$(date +%s%N)
timeout 600 python bench.py --detail $O/${TAG}_bench_detail_default.json > $O/${TAG}_bench_default.log 2>&1
t1=$(date +%s%N)
grep '^{' $O/${TAG}_bench_default.log > $O/${TAG}_bench_default_flags.json
echo "python bench.py (default flags): wall $(( (t1 - t0) / 1000000 )) ms, printed line $(wc -c < $O/${TAG}_bench_default_flags.json) bytes" > $O/${TAG}_bench_wall.txt
cat $O/${TAG}_bench_wall.txt
{
  PMC_TIMEOUT=300 timeout 1300 python tools/r06/stream_pmc.py cfg2b 16
  VARIANT=2:4 PMC_TIMEOUT=300 timeout 1300 python tools/r06/stream_pmc.py configs4 16
  VARIANT=1:3 PMC_TIMEOUT=300 timeout 1300 python tools/r06/stream_pmc.py configs3 16
} > $O/${TAG}_stream_pmc.txt 2>&1
cp gpurun_out/first_process/log.txt $O/${TAG}_first_process.txt 2>/dev/null
ls -la $O
