#!/bin/bash
# Round-4 evidence for bench.py's line.  Run on the GPU box from the repo root:
#   bash tools/prof_r04.sh [tag]  →  gpurun_out/profiles_<tag>/ (copy into profiles/)
#  1. rocprofv3 --kernel-trace --stats of the bench command — the primary workload alone
#     (kernel averages comparable with roofline.launch_ms_avg) and the full line;
#     no counters in these runs;
#  2. the plain bench.py run — it collects its own counters in separate `--pmc` passes
#     (bench.py:collect_counters) — whose JSON line is the committed result;
#  3. ingest / denoiser / cancel figures of the C++ drop-in test.
TAG=${1:-r04}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/profiles_$TAG
rm -rf $O; mkdir -p $O
cd /tmp
# (YTHIP_LPT_PROBE=0 for the traces: the first batch of a tile grid is otherwise launched as 1 + 63 samples —
#  bit-identical, but the 1-sample launch would sit in the kernel's average next to the full-size ones)
export YTHIP_LPT_PROBE=0
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- \
  python $R/bench.py --steps 3 --warmup 1 --no-counters --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_under_rocprof.log 2>&1
find /tmp/kt_$TAG -name '*_kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats.csv \;
grep '^{' $O/${TAG}_bench_under_rocprof.log > $O/${TAG}_bench_under_rocprof.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kta_$TAG -- \
  python $R/bench.py --steps 3 --warmup 1 --no-counters --no-cpu-baseline > /dev/null 2>&1
find /tmp/kta_$TAG -name '*_kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats_all.csv \;
unset YTHIP_LPT_PROBE
timeout 420 python $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.log 2>&1
grep '^{' $O/${TAG}_bench.log > $O/${TAG}_bench.json
cd $R
{
  echo "# oracle/_ref/dropin_test (sections 4b, 4c): ingest and denoiser"
  timeout 120 oracle/_ref/dropin_test 2>&1 | grep -E "ingest|denoiser|trace_cancel|dropin_test"
  echo "# tools/denoise_time.py"
  timeout 120 python tools/denoise_time.py 2>&1 | grep denoise
} > $O/${TAG}_ingest_denoise.txt 2>&1
ls -la $O
