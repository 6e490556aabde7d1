#!/bin/bash
# Developer tool (GPU box): seeded (strength, bleed) pairs on 8192-pixel strips, library variants side by side (name "base" = the tree's library): tools/gpu_r5_seeded_ab.sh TAG name ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $R
cp pngloss_amd/csrc/libpngloss_hip.so /tmp/keep.so
: > $OUT/${TAG}_seeded.txt
for V in "$@"; do
  if [ "$V" = base ]; then cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so; else cp tools/ablate_build/libpngloss_hip_$V.so pngloss_amd/csrc/libpngloss_hip.so; fi
  echo "=== $V" >> $OUT/${TAG}_seeded.txt
  for SB in "40 1" "85 1" "85 2" "160 1" "255 3"; do
    python tests/tools/gpu_seg_time.py ${SW:-8192} ${SH:-1024} 0 $SB 2 2>&1 | grep engine | sort -t= -k3 | tail -1 >> $OUT/${TAG}_seeded.txt
  done
done
cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "sweep_8192 or strengths_and_bleeds or every_candidate_count or seeded" 2>&1 | tail -3 ) >> $OUT/${TAG}_seeded.txt
