#!/bin/bash
# round 6, session 35: pathdirect on the streaming scheduler, shipped form — stream + bench-contract tests, A/B at the bench's batch sizes
cd "$(dirname "$0")/../.."
out=$PWD/gpurun_out/r06_gpu35; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | tail -n 12 > $out/pytest.txt; cat $out/pytest.txt
{
  date
  SAMPLER=pathdirect SCENES=cfg2b,cornell9m,configs4,configs3 VARIANTS=0:4 LAUNCHES=2 timeout 900 python tools/r06/stream_ab.py
  SAMPLER=pathdirect FASTMATH=2 SCENES=cfg2b,cornell9m VARIANTS=0:4 LAUNCHES=2 timeout 900 python tools/r06/stream_ab.py
  date
} > $out/stream_ab_pathdirect.txt 2>&1
cat $out/stream_ab_pathdirect.txt
