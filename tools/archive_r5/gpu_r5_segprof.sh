#!/bin/bash
# phase clocks of the segment engine's kernels (PNGLOSS_HIP_SEGPROF: the clocks slow the run down; proportions only) on an n-frame 1080p batch
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
N=${1:-32}; TAG=${2:-r05_segprof}
cd /tmp && export TMPDIR=/tmp
cd $R
PNGLOSS_HIP_ENGINE=seg PNGLOSS_HIP_SEGPROF=1 PNGLOSS_HIP_DEBUG=1 PNGLOSS_HIP_SEG_GROUPS=${G:-1} python /tmp/bn.py $N > $OUT/${TAG}_$N.txt 2>&1
