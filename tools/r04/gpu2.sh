#!/bin/bash
# Round 4, GPU session 2: traversal A/B (instance records in TLAS-leaf order, root-box pretest, cooperative sections),
# the tolerance-mode tests, and the vector-memory-path counters (TA / TCP) of three workloads.  -> gpurun_out/r04_gpu2/
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu2; mkdir -p $out
export TMPDIR=/tmp
ab() {  # ab SCENE RES SPP lib...
  local sc=$1 res=$2 spp=$3; shift 3
  for n in "$@"; do
    printf "%-10s " $n
    YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=path DIGEST=1 timeout 300 python tools/sampler_times.py 2>&1 | grep -v "^\[timing\]" | tail -n 1
  done
}
{
  date
  ab plane 1280 64 base2 tleaf pre cooptlas
  ab cornell1m 1024 16 base2 tleaf pre cooptlas
  ab cfg4 1920 32 base2 tleaf pre cooptlas coopboth
  ab cfg5 1280 16 base2 tleaf pre coopleaf coopboth
  ab instances 1280 64 base2 tleaf pre cooptlas coopboth
  ab lines_points 1280 64 base2 pre coopleaf coopboth
  ab materials 1280 64 base2 tleaf pre cooptlas
  date
} > $out/ab.txt 2>&1
{
  echo "== tolerance-mode tests"; date
  timeout 1200 python -m pytest -x -q -s tests/test_gpu_fastmath.py 2>&1 | grep -v "^$" | tail -40
  date
} > $out/fast.txt 2>&1
{
  echo "== vector memory path (TA / TCP) counters"; date
  for w in configs1 cfg2b configs3 configs4; do
    timeout 600 python tools/pmc_any.py $w GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum \
       -- GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum \
       -- GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum SQ_INSTS_VMEM_RD 2>&1 | tail -40
  done
  date
} > $out/pmc.txt 2>&1
cat $out/ab.txt $out/fast.txt $out/pmc.txt
