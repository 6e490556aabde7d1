#!/bin/bash
# round 6: kernel trace (rocprofv3 --kernel-trace --stats) of an n-frame 1080p batch on the segment engine, first n frames of configs[3], synchronous entry, launch groups opted in
# usage: [env hooks] tools/gpu_r6_prof.sh N TAG
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
N=${1:-32}
TAG=${2:-r06_prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
PNGLOSS_HIP_ENGINE=seg SHARE_REPS=1 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o trace --output-format csv -- python tests/tools/gpu_rank_share.py $N > $OUT/${TAG}_prof.log 2>&1
{ echo "# PNGLOSS_HIP_ENGINE=seg SEG_SEEDS=${PNGLOSS_HIP_SEG_SEEDS:-default} SEG_GROUPS=${PNGLOSS_HIP_SEG_GROUPS:-default} SEED_KIN=${PNGLOSS_HIP_SEED_KIN:-default} LIB=${PNGLOSS_HIP_LIBNAME:-default} rocprofv3 --kernel-trace --stats -- python tests/tools/gpu_rank_share.py $N  (one run of the first $N frames of configs[3])"
  grep "^n=" $OUT/${TAG}_prof.log
  cat $(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1) | cut -c1-220 | head -9; } > $OUT/${TAG}_kernel_stats_$N.txt
rm -rf $OUT/${TAG}_prof
