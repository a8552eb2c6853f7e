#!/bin/bash
# A/B of one development build under different environment settings (runtime options):
#   tools/ab_env.sh NAME "VAR=a" "VAR=b" ...     (SCENES / SAMPLERS as tools/ab_libs.sh)
name=$1; shift
for sc in ${SCENES:-plane cornell1m cfg4 cfg5}; do
  case $sc in cornell1m) export RES=1024 SPP=16;; cfg4) export RES=1920 SPP=32;; cfg5) export RES=1280 SPP=16;; *) export RES=1280 SPP=64;; esac
  for e in "$@"; do
    printf "%-22s " "$e"
    env $e YTHIP_LIB=$PWD/build/dev/libythip_$name.so SCENE=$sc SAMPLERS=${SAMPLERS:-path} DIGEST=1 timeout 300 python tools/sampler_times.py 2>&1 | grep -v "^\[timing\]" | tail -n 1
  done
done
