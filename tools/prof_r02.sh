#!/bin/bash
# Round-2 evidence for bench.py's line.  Run on the GPU box from the repo root:
#   bash tools/prof_r02.sh [tag]  →  gpurun_out/profiles_<tag>/ (copy into profiles/)
#  1. rocprofv3 --kernel-trace --stats of the bench command (no counters in this run);
#  2. the plain bench.py run — it collects its own counters in separate `--pmc` passes
#     (bench.py:collect_counters) — whose JSON line is the committed result.
TAG=${1:-r02}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/profiles_$TAG
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- \
  python $R/bench.py --steps 3 --warmup 1 --no-counters --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.log 2>&1
find /tmp/kt_$TAG -name '*_kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats.csv \;
grep '^{' $O/${TAG}_bench_under_rocprof.log > $O/${TAG}_bench_under_rocprof.json
python $R/bench.py --steps 3 --warmup 1 > $O/${TAG}_bench.log 2>&1
grep '^{' $O/${TAG}_bench.log > $O/${TAG}_bench.json
ls -la $O
