#!/bin/bash
# A/B of prebuilt libraries in tools/ablate_build/ on 1080p batches (segment engine, library defaults): tools/gpu_r5_ab.sh tag lib1 lib2 ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $R
cp pngloss_amd/csrc/libpngloss_hip.so /tmp/keep.so
: > $OUT/${TAG}_ab.txt
for L in base "$@"; do
  [ $L = base ] && cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so || cp tools/ablate_build/libpngloss_hip_$L.so pngloss_amd/csrc/libpngloss_hip.so
  echo "=== $L" >> $OUT/${TAG}_ab.txt
  SEG_BATCH_ENGINES=seg timeout 600 python tests/tools/gpu_seg_batch.py 1920 1080 ${NS:-16 32 64} 2>&1 | grep "n=" >> $OUT/${TAG}_ab.txt
done
cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so
