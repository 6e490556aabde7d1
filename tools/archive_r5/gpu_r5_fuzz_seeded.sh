#!/bin/bash
# Developer tool (GPU box): parity campaign pinned to the segment engine at strengths whose state sets are SEEDED (the gather kernel's path), small and large shapes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
: > gpurun_out/r05f_fuzz_seeded.txt
for S in 85 160 255 48; do
  for MODE in "" big; do
    echo "=== FUZZ_STRENGTH=$S FUZZ_ENGINES=seg $MODE" >> gpurun_out/r05f_fuzz_seeded.txt
    FUZZ_STRENGTH=$S FUZZ_ENGINES=seg timeout 400 python tests/tools/gpu_fuzz.py ${FUZZ_SECS:-70} $((S * 7 + 1)) $MODE 2>&1 | tail -2 >> gpurun_out/r05f_fuzz_seeded.txt
  done
done
