#!/bin/bash
# Stall attribution of k_trace from hardware counters (PC sampling and ATT are
# not available on this pool — `rocprofv3 -L` lists no PC-sampling configuration for gfx950 and
# the ATT decoder is not installed — so the attribution comes from SQ / TCP / TCC counters,
# collected in separate --pmc passes, counters only; a pass that asks for more than the hardware can
# collect at once aborts inside rocprofv3 and then HANGS — hence the short per-pass timeout).
#   tools/stall_attr.sh SCENE [LIBNAME]   -> gpurun_out/stall/<scene>[_lib].txt
export TMPDIR=/tmp
R=$PWD; SC=${1:-cornell1m}; LIB=${2:-}
case $SC in cornell1m|cornell9m) export RES=1024 SPP=16;; cfg4) export RES=1920 SPP=32;; cfg5) export RES=1280 SPP=16;; *) export RES=1280 SPP=64;; esac
[ -n "$LIB" ] && export YTHIP_LIB=$R/build/dev/libythip_$LIB.so
tag=$SC${LIB:+_$LIB}
O=$R/gpurun_out/stall; mkdir -p $O; cd /tmp
run() { n=$1; shift
  SCENE=$SC SAMPLERS=path timeout -k 5 45 rocprofv3 --pmc "$@" --kernel-include-regex k_trace --output-format csv -d $O/${tag}_$n -- python $R/tools/sampler_times.py > $O/${tag}_$n.log 2>&1 || echo "pass $n failed: $(tail -n 1 $O/${tag}_$n.log)"
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES
run p2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS
run p3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
run p4 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_VMEM_RD
run p5 TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
run p5b TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
run p6 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum
run p7 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
run p7b TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum
run p8 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64
run p9 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU
python3 - <<PY > $O/$tag.txt
import csv,glob,collections
tot={}
for p in ["p1","p2","p3","p4","p5","p5b","p6","p7","p7b","p8","p9"]:
    acc=collections.defaultdict(float); n=collections.defaultdict(set); dur=[]
    for f in glob.glob("$O/${tag}_"+p+"/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_trace" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    for c,x in sorted(acc.items()):
        tot[c]=x/len(n[c]); print("%-4s %-36s %.6g per launch (%d launches)"%(p,c,tot[c],len(n[c])))
g=tot.get
def ratio(a,b,txt,scale=1.0):
    if g(a) and g(b): print("%-64s %.4g"%(txt,scale*g(a)/g(b)))
print()
ratio("SQ_WAIT_ANY","SQ_WAVE_CYCLES","wave time waiting at s_waitcnt (any)")
ratio("SQ_WAIT_INST_ANY","SQ_WAVE_CYCLES","wave time waiting for an instruction to issue (any)")
ratio("SQ_WAIT_INST_LDS","SQ_WAVE_CYCLES","wave time waiting for an LDS instruction to issue")
ratio("SQ_ACTIVE_INST_ANY","SQ_WAVE_CYCLES","wave time with an instruction executing")
ratio("SQ_THREAD_CYCLES_VALU","SQ_ACTIVE_INST_VALU","active lanes per VALU instruction-cycle (of 64)")
ratio("SQ_INST_LEVEL_VMEM","SQ_INSTS_VMEM_RD","mean cycles a VMEM read instruction is in flight")
ratio("TCP_TCC_READ_REQ_LATENCY_sum","TCP_TCC_READ_REQ_sum","mean L1->L2 read latency (cycles)")
ratio("TCP_TCP_LATENCY_sum","TCP_TOTAL_ACCESSES_sum","mean L1 access latency (cycles)")
ratio("TCC_EA0_RDREQ_LEVEL_sum","TCC_EA0_RDREQ_sum","mean L2->fabric read latency (cycles)")
ratio("TCP_TCC_READ_REQ_sum","TCP_TOTAL_CACHE_ACCESSES_sum","L1 miss rate (read requests to L2 / cache accesses)")
ratio("TCC_HIT_sum","TCC_REQ_sum","L2 hit rate")
ratio("TCP_UTCL1_TRANSLATION_MISS_sum","TCP_UTCL1_REQUEST_sum","UTCL1 (TLB) miss rate")
ratio("SQ_INSTS_VALU","SQ_INSTS","VALU share of all instructions")
PY
cat $O/$tag.txt
