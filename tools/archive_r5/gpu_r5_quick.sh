#!/bin/bash
# quick look: 1080p batches on the segment engine at the library's defaults + a G=1 kernel trace of 32 frames
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05_q}
cd /tmp && export TMPDIR=/tmp
cd $R
echo "=== $TAG (library defaults)" > $OUT/${TAG}_batch.txt
SEG_BATCH_ENGINES=seg timeout 600 python tests/tools/gpu_seg_batch.py 1920 1080 ${NS:-8 16 32 64} >> $OUT/${TAG}_batch.txt 2>&1
PNGLOSS_HIP_SEG_GROUPS=1 bash tools/gpu_r5_prof.sh 32 $TAG
