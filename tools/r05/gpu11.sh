#!/bin/bash
# Round 5, GPU session 11: can bench.py's three counter passes be two?  (A pass that asks for too much has been seen to hang: short timeouts.)
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu11
out=gpurun_out/r05_gpu11; mkdir -p $out
export TMPDIR=/tmp
{
  echo "== merged second pass"
  PMC_TIMEOUT=60 timeout 120 python tools/pmc_any.py cfg2b WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE 2>&1 | grep -v "^\[bench\]"
  echo "== the three passes of bench.py, for comparison"
  PMC_TIMEOUT=60 timeout 200 python tools/pmc_any.py cfg2b WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU -- TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE 2>&1 | grep -v "^\[bench\]"
  echo "== everything in ONE pass"
  PMC_TIMEOUT=60 timeout 120 python tools/pmc_any.py cfg2b FETCH_SIZE TCC_REQ_sum SQ_INSTS_VALU SQ_WAVE_CYCLES WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE 2>&1 | grep -v "^\[bench\]"
} > $out/pmc.txt 2>&1
cat $out/pmc.txt
