#!/bin/bash
# What bounds k_trace on the incoherent BASELINE scenes?  One PMC pass (SQ counters only,
# no tracing) per scene over tools/sampler_times.py's `path` run.
#   tools/pmc_scenes.sh [scene ...]   -> gpurun_out/pmc_scenes/<scene>.txt
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_scenes; rm -rf $O; mkdir -p $O; cd /tmp
for sc in ${@:-cornell1m cfg4}; do
  case $sc in cornell1m) export RES=1024 SPP=16;; cfg4) export RES=1280 SPP=32;; *) export RES=1280 SPP=64;; esac
  SCENE=$sc SAMPLERS=path rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES \
    --output-format csv -d $O/$sc -- python $R/tools/sampler_times.py > $O/$sc.log 2>&1
  SCENE=$sc SAMPLERS=path rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/${sc}_tcc -- python $R/tools/sampler_times.py > $O/${sc}_tcc.log 2>&1
  python - <<PY > $O/$sc.txt
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for d in ("$O/$sc", "$O/${sc}_tcc"):
  for f in glob.glob(d+"/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "k_trace" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])].add(r["Dispatch_Id"])
for k in acc:
    print("$sc", k)
    v={c: x/len(n[(k,c)]) for c,x in acc[k].items()}
    for c,x in sorted(v.items()): print("   %-24s %.4g per dispatch"%(c, x))
    if "GRBM_GUI_ACTIVE" in v and "SQ_ACTIVE_INST_VALU" in v:
        cyc=v["GRBM_GUI_ACTIVE"]/8
        print("   valu_busy = SQ_ACTIVE_INST_VALU*4/(cycles*1024) = %.3f"%(v["SQ_ACTIVE_INST_VALU"]*4/(cyc*1024)))
        print("   wave wait share = SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES = %.3f"%(v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"]))
    if "TCC_REQ_sum" in v: print("   L2 hit rate = %.3f"%(v["TCC_HIT_sum"]/v["TCC_REQ_sum"]))
PY
  cat $O/$sc.txt
done
