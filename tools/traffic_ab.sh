#!/bin/bash
# HBM-side traffic + time of `path` on a scene for development builds: one FETCH_SIZE and one
# WRITE_SIZE pass each (counters only).   tools/traffic_ab.sh SCENE NAME [NAME ...]
export TMPDIR=/tmp
R=$PWD; sc=$1; shift
case $sc in cornell1m) export RES=1024 SPP=16;; cfg4) export RES=1920 SPP=32;; cfg5) export RES=1280 SPP=16;; *) export RES=1280 SPP=64;; esac
for n in "$@"; do
  export YTHIP_LIB=$R/build/dev/libythip_$n.so
  t=$(cd $R; SCENE=$sc SAMPLERS=path python tools/sampler_times.py 2>&1 | tail -n 1)
  for c in FETCH_SIZE WRITE_SIZE; do
    O=/tmp/traf_$n_$c; rm -rf $O
    (cd /tmp; SCENE=$sc SAMPLERS=path rocprofv3 --pmc $c --output-format csv -d $O -- python $R/tools/sampler_times.py > /dev/null 2>&1)
    python3 - <<PY
import csv,glob
v=[]; 
for f in glob.glob("$O/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]: v.append(float(r["Counter_Value"]))
# per dispatch: rows are per (dispatch, xcd?) -> sum per dispatch id
import collections
d=collections.defaultdict(float)
for f in glob.glob("$O/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]: d[r["Dispatch_Id"]]+=float(r["Counter_Value"])
m=sum(d.values())/max(len(d),1)
print("   %-10s %-10s %.3f GB per launch%s"%("$n","$c", m*1024/1e9*(2 if "$c"=="FETCH_SIZE" else 1), " (x2 gfx950 correction applied)" if "$c"=="FETCH_SIZE" else ""))
PY
  done
  echo "$n $t"
done
