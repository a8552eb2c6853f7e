#!/bin/bash
# A/B of development builds (tools/devbuild.sh NAME ...): times + whole-state digests of `path`
# on the BASELINE scenes.   tools/ab_libs.sh NAME [NAME ...]
for sc in ${SCENES:-plane cornell1m cfg4 cfg5}; do
  case $sc in cornell1m|cornell9m) export RES=1024 SPP=16;; cfg4) export RES=1920 SPP=32;; cfg5) export RES=1280 SPP=16;; *) export RES=1280 SPP=64;; esac
  for n in "$@"; do
    printf "%-10s " $n
    YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc SAMPLERS=${SAMPLERS:-path} DIGEST=1 timeout 300 python tools/sampler_times.py 2>&1 | grep -v "^\[timing\]" | tail -n 2 | tr '\n' ' '; echo
  done
done
