#!/bin/bash
# Round 5, GPU session 4: why the hair's own-walk records differ in 0.4 % of the rays; counters of the own-tree kernels.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh gpu4
out=gpurun_out/r05_gpu4; mkdir -p $out
export TMPDIR=/tmp
{ SCENE=cfg5 timeout 300 python tools/r05/own_diag.py; SCENE=lines_points timeout 300 python tools/r05/own_diag.py; } > $out/diag.txt 2>&1
{
  for w in cfg2b cfg2b:own configs3 configs3:own configs4 configs4:own features1 features1:own configs1:own; do
    PMC_TIMEOUT=120 timeout 600 python tools/pmc_any.py $w FETCH_SIZE TCC_REQ_sum SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- \
       WRITE_SIZE SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU -- \
       TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE 2>&1 | grep -v "^\[bench\]"
  done
} > $out/pmc.txt 2>&1
cat $out/diag.txt $out/pmc.txt
