#!/bin/bash
# round 6, session 3: compiler scheduling flags (never explored before: docs/HISTORY.md has the SLP vectorizer only) — development builds
# of tools/devbuild.sh, `path`, whole-state digests must agree along a row of builds
out=gpurun_out/r06_gpu20; mkdir -p $out
{
  date
  SCENES="plane cornell1m cfg4 cfg5 corpus:features1" bash tools/ab_libs.sh $VARIANTS
  SCENES="plane cornell1m cfg4 cfg5 corpus:features1" bash tools/ab_libs.sh $VARIANTS
  date
} > $out/${ABNAME:-flags_ab}.txt 2>&1
cat $out/${ABNAME:-flags_ab}.txt
