#!/bin/bash
# PMC comparison of k_trace and k_pool on one scene (separate passes, counters only).
#   tools/pmc_ab.sh SCENE [POOL_CFG]   -> gpurun_out/pmc_ab/<scene>_<pass>.txt
export TMPDIR=/tmp
R=$PWD; SC=${1:-cfg4}; CFG=${2:-0:128:16:64:1}
O=$R/gpurun_out/pmc_ab; mkdir -p $O; cd /tmp
run() { # name, counters...
  n=$1; shift
  SCENES=$SC STEPS=1 POOL_CFGS=$CFG timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/${SC}_$n -- python $R/tools/pool_ab.py > $O/${SC}_$n.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
#run sq2 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
#run tcp2 TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - <<PY > $O/$SC.txt
import csv,glob,collections,sys
for p in ("sq1","sq2","tcp1","tcp2","tcc"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for f in glob.glob("$O/${SC}_"+p+"/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if "k_trace" not in k and "k_pool" not in k: continue
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])].add(r["Dispatch_Id"])
    for k in acc:
        for c,x in sorted(acc[k].items()): print("%-6s %-48s %-32s %.4g per dispatch"%(p,k[:48],c,x/len(n[(k,c)])))
PY
cat $O/$SC.txt
