#!/bin/bash
# A/B development builds of libythip (path / pathtest / naive kernels only, ~20 s):
#   tools/devbuild.sh NAME [-DFLAG ...]   → build/dev/libythip_NAME.so
# run with  YTHIP_LIB=build/dev/libythip_NAME.so python tools/sampler_times.py
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build/dev
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -ffp-contract=off -fPIC -DYT_DEV_ONLY_PATH "$@" \
  -c -o build/dev/ythip_$name.o yocto-gl_amd/csrc/ythip.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -ffp-contract=off -fPIC -c -o build/dev/gpubuild_$name.o yocto-gl_amd/csrc/yt_gpubuild.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c -o build/dev/multi_$name.o yocto-gl_amd/csrc/yt_multi.hip
[ -f build/dev/order.o ] && [ build/dev/order.o -nt yocto-gl_amd/csrc/yt_order.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c -o build/dev/order.o yocto-gl_amd/csrc/yt_order.hip
[ -f build/dev/io.o ] && [ build/dev/io.o -nt yocto-gl_amd/csrc/yt_io.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c -o build/dev/io.o yocto-gl_amd/csrc/yt_io.hip
[ -f build/dev/sceneio.o ] && [ build/dev/sceneio.o -nt yocto-gl_amd/csrc/yt_sceneio.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c -o build/dev/sceneio.o yocto-gl_amd/csrc/yt_sceneio.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/dev/libythip_$name.so build/dev/ythip_$name.o build/dev/gpubuild_$name.o build/dev/multi_$name.o build/dev/order.o build/dev/io.o build/dev/sceneio.o -ldl -lz
rm -f build/dev/ythip_$name.o build/dev/gpubuild_$name.o build/dev/multi_$name.o
echo built build/dev/libythip_$name.so
