#!/bin/bash
# rocprofv3 evidence for bench.py's numbers.  Run on the GPU box from the repo
# root:  bash tools/prof.sh [tag]      → gpurun_out/prof_<tag>/ (+ summaries that
# tools/prof_collect.py copies into profiles/).
# Kernel timing and PMC counters are collected in SEPARATE runs (never --pmc
# together with tracing), one counter group per run (TCC has 4 slots).
TAG=${1:-r01}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $BENCH > $O/kt.log 2>&1
B="python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-roofline"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_sq1 -- $B > $O/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_tcc -- $B > $O/pmc_tcc.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_tcp -- $B > $O/pmc_tcp.log 2>&1
# calibration of FETCH_SIZE / WRITE_SIZE on a known byte count (MI355X_MICROARCH.md §HBM)
python $R/tools/prof_collect.py $O $TAG > $O/collect.log 2>&1
du -sh $O
