#!/bin/bash
# per-sampler instruction mix of k_trace on configs[1] (diagnostic)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_samplers; rm -rf $O; mkdir -p $O; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/p1 -- python $R/tools/sampler_times.py > $O/p1.log 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob("$O/p1/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "k_trace" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(n[k]))
    for c,v in sorted(acc[k].items()): print("   %-24s %.4g per dispatch"%(c, v/len(n[k])))
PY
