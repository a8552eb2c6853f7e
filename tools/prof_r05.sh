#!/bin/bash
# Round-5 evidence for bench.py's line.  Run on the GPU box from the repo root:
#   bash tools/prof_r05.sh [tag]  →  gpurun_out/profiles_<tag>/ (copy into profiles/)
#  1. rocprofv3 --kernel-trace --stats of the bench command — the primary workload alone (kernel averages comparable with
#     roofline.launch_ms_avg) and the full line (every mode's kernels); no counters in these runs;
#  2. the plain bench.py run as the driver runs it (default flags) — it collects its own counters in separate `--pmc` passes
#     (bench.py: collect_counters): the printed line (< 4 KB) and the side file bench_detail.json are the committed result; and once
#     more with --steps 20 --warmup 5 --tolerance-counters (every mode's fractions);
#  3. the instruction budgets, the own-tree gates, ingest / denoiser figures of the C++ drop-in test.
TAG=${1:-r05}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/profiles_$TAG
rm -rf $O; mkdir -p $O
cd /tmp
export YTHIP_LPT_PROBE=0  # (the traces: no 1 + 63-sample split of a first batch in the kernel's average)
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- \
  python $R/bench.py --steps 3 --warmup 1 --no-counters --no-cpu-baseline --no-other-configs --detail /tmp/detail_a.json > $O/${TAG}_bench_under_rocprof.log 2>&1
find /tmp/kt_$TAG -name '*_kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats.csv \;
grep '^{' $O/${TAG}_bench_under_rocprof.log > $O/${TAG}_bench_under_rocprof.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kta_$TAG -- \
  python $R/bench.py --steps 3 --warmup 1 --no-counters --no-cpu-baseline --detail /tmp/detail_b.json > /dev/null 2>&1
find /tmp/kta_$TAG -name '*_kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats_all.csv \;
unset YTHIP_LPT_PROBE
cd $R
t0=$(date +%s%N)
timeout 600 python bench.py --detail $O/${TAG}_bench_detail_default.json > $O/${TAG}_bench_default.log 2>&1
t1=$(date +%s%N)
grep '^{' $O/${TAG}_bench_default.log > $O/${TAG}_bench_line_default.json
echo "python bench.py (default flags): wall $(( (t1 - t0) / 1000000 )) ms, printed line $(wc -c < $O/${TAG}_bench_line_default.json) bytes" > $O/${TAG}_bench_wall.txt
timeout 900 python bench.py --steps 20 --warmup 5 --tolerance-counters --detail $O/${TAG}_bench_detail.json > $O/${TAG}_bench.log 2>&1
grep '^{' $O/${TAG}_bench.log > $O/${TAG}_bench_line.json
{
  timeout 300 python tools/instruction_budget.py plane 1280 64
  echo
  timeout 300 python tools/instruction_budget.py cornell1m 1024 16
} > $O/${TAG}_instruction_budget.txt 2>&1
[ -n "$SKIP_OWN_GATES" ] || timeout 1200 python -m pytest -q -s tests/test_gpu_own_tree.py 2>&1 | grep "hits\]\|gate\]\|passed\|failed" | sed 's/^\.*//' > $O/${TAG}_own_tree_gates.txt
{
  echo "# oracle/_ref/dropin_test (sections 4b, 4c): ingest and denoiser"
  timeout 120 oracle/_ref/dropin_test 2>&1 | grep -E "ingest|denoiser|trace_cancel|dropin_test"
  echo "# tools/denoise_time.py"
  timeout 120 python tools/denoise_time.py 2>&1 | grep denoise
} > $O/${TAG}_ingest_denoise.txt 2>&1
cp gpurun_out/first_process/log.txt $O/${TAG}_first_process.txt 2>/dev/null
ls -la $O
