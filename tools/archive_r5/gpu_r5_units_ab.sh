#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
cp pngloss_amd/csrc/libpngloss_hip.so /tmp/keep.so
: > $OUT/r05_units_ab.txt
for U in 2 3; do
  cp tools/ablate_build/libpngloss_hip_u$U.so pngloss_amd/csrc/libpngloss_hip.so
  for G in 2 3; do
  echo "=== SEG_UNIT=$U GROUPS=$G" >> $OUT/r05_units_ab.txt
  PNGLOSS_HIP_SEG_UNIT=1 PNGLOSS_HIP_SEG_GROUPS=$G SEG_BATCH_ENGINES=seg timeout 600 python tests/tools/gpu_seg_batch.py 1920 1080 8 16 32 64 >> $OUT/r05_units_ab.txt 2>&1
  done
done
cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so
echo "=== SEG_UNIT=4 GROUPS=3" >> $OUT/r05_units_ab.txt
PNGLOSS_HIP_SEG_UNIT=1 PNGLOSS_HIP_SEG_GROUPS=3 SEG_BATCH_ENGINES=seg timeout 600 python tests/tools/gpu_seg_batch.py 1920 1080 8 16 32 64 >> $OUT/r05_units_ab.txt 2>&1
