#!/bin/bash
# round 6, session 3: the streaming scheduler in the tolerance / own-tree modes (yt_stream_unit.h) — tests, then the A/B
out=gpurun_out/r06_gpu16; mkdir -p $out
python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py -m gpu -x -q -k "stream or dropin" > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
for fm in 2 1; do
  FASTMATH=$fm SCENES=cfg2b,cornell9m,configs3,configs4,materials1 VARIANTS=1:3 LAUNCHES=2 timeout 900 python tools/r06/stream_ab.py 2>&1 | tee -a $out/stream_ab_modes.txt
done
