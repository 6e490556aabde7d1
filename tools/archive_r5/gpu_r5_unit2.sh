#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05_unit2}
cd /tmp && export TMPDIR=/tmp
cd $R
echo "=== $TAG PNGLOSS_HIP_SEG_UNIT=1" > $OUT/${TAG}_batch.txt
PNGLOSS_HIP_SEG_UNIT=1 SEG_BATCH_ENGINES=seg timeout 600 python tests/tools/gpu_seg_batch.py 1920 1080 4 8 16 32 64 128 >> $OUT/${TAG}_batch.txt 2>&1
PNGLOSS_HIP_SEG_UNIT=1 bash tools/gpu_r5_prof.sh 32 $TAG
