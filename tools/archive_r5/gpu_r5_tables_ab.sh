#!/bin/bash
# Developer tool (GPU box): decision tables of wide bands built in chunks (base = the tree's library) against the plain scan (tools/ablate_build/libpngloss_hip_old.so):
# seeded and chunked (strength, bleed) pairs on 8192-pixel strips, the headline strip, a kernel trace of s=85 b=2 with either library.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05t}
cd /tmp && export TMPDIR=/tmp
cd $R
cp pngloss_amd/csrc/libpngloss_hip.so /tmp/keep.so
: > $OUT/${TAG}_tables.txt
for V in ${VARIANTS:-old base old base}; do
  if [ "$V" = base ]; then cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so; else cp tools/ablate_build/libpngloss_hip_$V.so pngloss_amd/csrc/libpngloss_hip.so; fi
  echo "=== $V" >> $OUT/${TAG}_tables.txt
  for SB in "40 1" "40 2" "85 1" "85 2" "85 8" "160 1" "255 3"; do
    python tests/tools/gpu_seg_time.py 8192 1024 0 $SB 2 2>&1 | grep engine | sort -t= -k3 | tail -1 >> $OUT/${TAG}_tables.txt
  done
done
cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so
AB_ROUNDS=${AB_ROUNDS:-4} bash tools/gpu_r5_ab_head.sh ${TAG} ${HEADVARS:-old}
cat $OUT/${TAG}_abhead.txt >> $OUT/${TAG}_tables.txt
for V in ${TRACEVARS:-old base}; do
  if [ "$V" = base ]; then cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so; else cp tools/ablate_build/libpngloss_hip_$V.so pngloss_amd/csrc/libpngloss_hip.so; fi
  PNGLOSS_HIP_ENGINE=seg rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_$V -o trace --output-format csv -- python tests/tools/gpu_seg_time.py 8192 1024 0 85 2 1 > /dev/null 2>&1
  { echo "=== kernel trace, s=85 b=2, $V"; head -7 $(find $OUT/${TAG}_prof_$V -name "*kernel_stats.csv" | head -1) | cut -c1-160; } >> $OUT/${TAG}_tables.txt
  rm -rf $OUT/${TAG}_prof_$V
done
cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "sweep_8192 or strengths_and_bleeds or every_candidate_count or seeded" 2>&1 | tail -3 ) >> $OUT/${TAG}_tables.txt
