#!/bin/bash
# Round 6, GPU session 9: the adopted pair of walk leads (seven-load records + leaf-record shading where the mesh has no vertex
# normals) against round 5's walk and against leaf shading alone; the NEE samplers by scene class; bench contract + stream tests.
cd "$(dirname "$0")/../.."
bash tools/r05/session_head.sh r06_gpu9
out=$PWD/gpurun_out/r06_gpu9; mkdir -p $out
export TMPDIR=/tmp
{
  date
  SCENES="plane cornell1m cfg4 cfg5 cornell9m corpus:features1" bash tools/ab_libs.sh r6alloff r6b r6b_w8 r6alloff r6b r6b_w8
  echo "--- pathdirect / pathmis: general class (YTHIP specialization off) vs scene class, product build"
  for spec in 0 1; do
    for sc in cornell1m cornellbox corpus:materials1; do
      printf "specialize %s  " $spec
      SPECIALIZE=$spec SCENE=$sc RES=$([ $sc = cornellbox ] && echo 512 || echo 1024) SPP=16 SAMPLERS=pathdirect,pathmis DIGEST=1 timeout 300 python tools/sampler_times.py 2>&1 | grep -v "^\[" | tr '\n' ' '; echo
    done
  done
  date
} > $out/ab.txt 2>&1
cat $out/ab.txt
timeout 1800 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_stream.py -x -q 2>&1 | tail -n 15 > $out/pytest.txt; cat $out/pytest.txt
