#!/bin/bash
# round 5: enumeration in units on the GPU: digests of batches + batch timing with and without it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
for U in 0 1; do
  echo "=== PNGLOSS_HIP_SEG_UNIT=$U" >> $OUT/r05_unit_batch.txt
  PNGLOSS_HIP_SEG_UNIT=$U SEG_BATCH_ENGINES=seg timeout 600 python tests/tools/gpu_seg_batch.py 1920 1080 1 4 8 16 32 64 >> $OUT/r05_unit_batch.txt 2>&1
done
PNGLOSS_HIP_ENGINE=seg PNGLOSS_HIP_SEG_UNIT=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "configs3_rank or golden_synthetic or suite_batch or segment_engine or edge_shapes or 1080p or headline" > $OUT/r05_unit_tests.txt 2>&1; echo "pytest rc=$?" >> $OUT/r05_unit_tests.txt
