#!/bin/bash
# A/B development builds of libythip (path / pathtest / naive kernels only, about a minute on 8 cores):
#   tools/devbuild.sh NAME [-DFLAG ...]   → build/dev/libythip_NAME.so
# run with  YTHIP_LIB=build/dev/libythip_NAME.so python tools/sampler_times.py
# The flags go to the units that compile device code from yt_kernels.h / yt_bvh.h (ythip.hip, yt_trace_*.hip, yt_fast.hip, yt_owntree.hip — DScene must
# have ONE layout across them); the other
# units are built once per source state (build/dev/*.o, rebuilt when their source or any header is newer) and shared.
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build/dev
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -ffp-contract=off -fPIC"
C=yocto-gl_amd/csrc
newest_header=$(ls -t $C/*.h include/ythip.h | head -1)
for u in yt_gpubuild yt_multi yt_order yt_io yt_sceneio; do
  o=build/dev/$u.o
  if [ ! -f $o ] || [ $C/$u.hip -nt $o ] || [ $newest_header -nt $o ]; then
    ( flock 9; if [ ! -f $o ] || [ $C/$u.hip -nt $o ] || [ $newest_header -nt $o ]; then $HIPCC $FLAGS -c -o $o.tmp.$$ $C/$u.hip && mv $o.tmp.$$ $o; fi ) 9> build/dev/.lock.$u &
  fi
done
pids=""
for u in ythip yt_bake yt_trace_path yt_trace_nee yt_trace_nee_cls yt_trace_misc yt_fast yt_owntree yt_stream; do
  $HIPCC $FLAGS -DYT_DEV_ONLY_PATH "$@" -c -o build/dev/${u}_$name.o $C/$u.hip & pids="$pids $!"
done
for p in $pids; do wait $p; done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o build/dev/libythip_$name.so build/dev/ythip_$name.o build/dev/yt_trace_path_$name.o \
  build/dev/yt_trace_nee_$name.o build/dev/yt_trace_nee_cls_$name.o build/dev/yt_stream_$name.o build/dev/yt_trace_misc_$name.o build/dev/yt_bake_$name.o build/dev/yt_gpubuild.o build/dev/yt_multi.o \
  build/dev/yt_order.o build/dev/yt_io.o build/dev/yt_sceneio.o build/dev/yt_fast_$name.o build/dev/yt_owntree_$name.o -ldl -lz
rm -f build/dev/ythip_$name.o build/dev/yt_stream_$name.o build/dev/yt_trace_*_$name.o build/dev/yt_bake_$name.o build/dev/yt_fast_$name.o build/dev/yt_owntree_$name.o
echo built build/dev/libythip_$name.so
