#!/bin/bash
# Round 4, GPU session 5: the three-level (8-slot, 256-B record) walk against the shipped two-level one; the vector-memory-path
# counters, one small pass at a time under a short timeout.  -> gpurun_out/r04_gpu5/
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu5; mkdir -p $out
export TMPDIR=/tmp
ab() {  # ab SCENE RES SPP lib...
  local sc=$1 res=$2 spp=$3; shift 3
  for n in "$@"; do
    printf "%-10s " $n
    YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=path DIGEST=1 timeout 90 python tools/sampler_times.py 2>&1 | grep -v "^\[timing\]" | tail -n 1
  done
}
{
  date
  ab plane 1280 64 base4 wide8
  ab cornell1m 1024 16 base4 wide8
  ab cfg4 1920 32 base4 wide8
  ab cfg5 1280 16 base4 wide8
  ab cornell9m 1024 16 base4 wide8
  ab instances 1280 64 base4 wide8
  ab lines_points 1280 64 base4 wide8
  ab materials 1280 64 base4 wide8
  ab cornellbox 1280 64 base4 wide8
  ab corpus:materials1 1280 64 base4 wide8
  date
} > $out/ab.txt 2>&1
{
  echo "== vector memory path (TA / TCP) counters, one small pass each, 75 s limit per pass"; date
  for w in configs1 cfg2b configs3; do
    for pass in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" "GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
                "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
      PMC_TIMEOUT=75 timeout 120 python tools/pmc_any.py $w $pass 2>&1 | grep -v "^\[bench\]" | tail -8
    done
  done
  date
} > $out/pmc.txt 2>&1
cat $out/ab.txt $out/pmc.txt
