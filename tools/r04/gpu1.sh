#!/bin/bash
# Round 4, GPU session 1: (a) the GPU tests that cover the ADVICE r3 fixes, (b) the traversal A/B of the instance-entry
# and cooperative-leaf builds (digests must be equal within a scene).  Output -> gpurun_out/r04_gpu1/
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_gpu1; mkdir -p $out
export TMPDIR=/tmp
{
  echo "== tests of the fixed paths"; date
  timeout 900 python -m pytest -x -q tests/test_gpu_pixel_pool.py tests/test_gpu_refit.py \
      "tests/test_gpu_build.py::test_device_instance_tree_renders_refits_and_downloads" \
      "tests/test_gpu_build.py::test_small_shapes_are_built_by_a_host_thread_pool" \
      tests/test_gpu_transfers.py -k "not fresh" 2>&1 | tail -15
  echo "== dropin"; timeout 600 oracle/_ref/dropin_test 2>&1 | tail -15
} > $out/tests.txt 2>&1
ab() {  # ab SCENE RES SPP lib...
  local sc=$1 res=$2 spp=$3; shift 3
  for n in "$@"; do
    printf "%-10s " $n
    YTHIP_LIB=$PWD/build/dev/libythip_$n.so SCENE=$sc RES=$res SPP=$spp SAMPLERS=path DIGEST=1 timeout 300 python tools/sampler_times.py 2>&1 | grep -v "^\[timing\]" | tail -n 1
  done
}
{
  date
  ab plane 1280 64 base2 tleaf pre cooptlas
  ab cornell1m 1024 16 base2 tleaf pre cooptlas
  ab cfg4 1920 32 base2 tleaf pre cooptlas coopboth
  ab cfg5 1280 16 base2 tleaf pre coopleaf coopboth
  ab instances 1280 64 base2 tleaf pre cooptlas coopboth
  ab lines_points 1280 64 base2 pre coopleaf coopboth
  ab materials 1280 64 base2 tleaf pre cooptlas
  date
} > $out/ab.txt 2>&1
{
  echo "== tolerance mode (main library): exact vs fastmath, ms/step"; date
  for spec in "plane 1280 64" "cornell1m 1024 16" "cfg4 1920 32" "cfg5 1280 16" "cornell9m 1024 16" "materials 1280 64"; do
    set -- $spec
    for fm in 0 1; do
      printf "fastmath=%d " $fm
      FASTMATH=$fm SCENE=$1 RES=$2 SPP=$3 SAMPLERS=path DIGEST=1 timeout 300 python tools/sampler_times.py 2>&1 | tail -n 1
    done
  done
  echo "== tolerance-mode tests"; date
  timeout 1200 python -m pytest -x -q -s tests/test_gpu_fastmath.py 2>&1 | grep -v "^$" | tail -40
  date
} > $out/fast.txt 2>&1
cat $out/tests.txt $out/ab.txt $out/fast.txt
