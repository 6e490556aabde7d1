#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
: > $OUT/r05_groups_batch.txt
for U in 1 0; do for G in 1 2 4; do
  echo "=== UNIT=$U GROUPS=$G" >> $OUT/r05_groups_batch.txt
  PNGLOSS_HIP_SEG_UNIT=$U PNGLOSS_HIP_SEG_GROUPS=$G SEG_BATCH_ENGINES=seg timeout 600 python tests/tools/gpu_seg_batch.py 1920 1080 8 16 32 64 128 >> $OUT/r05_groups_batch.txt 2>&1
done; done
PNGLOSS_HIP_ENGINE=seg PNGLOSS_HIP_SEG_UNIT=1 PNGLOSS_HIP_SEG_GROUPS=4 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "configs3_rank or golden_synthetic or suite_batch or segment_engine or edge_shapes or 1080p or mixed" > $OUT/r05_groups_tests.txt 2>&1; echo "pytest rc=$?" >> $OUT/r05_groups_tests.txt
