#!/bin/bash
# A/B of prebuilt libraries on the HEADLINE frame (interleaved runs on the same box; min and median of the engine ms): tools/gpu_r5_ab_head.sh tag lib...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $R
cp pngloss_amd/csrc/libpngloss_hip.so /tmp/keep.so
for L in base "$@"; do rm -f /tmp/ab_$L.txt; done
for i in $(seq 1 ${AB_ROUNDS:-5}); do
  for L in base "$@"; do
    [ $L = base ] && cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so || cp tools/ablate_build/libpngloss_hip_$L.so pngloss_amd/csrc/libpngloss_hip.so
    python tests/tools/gpu_seg_time.py ${AB_W:-4096} ${AB_H:-2048} 0 19 2 2 2>/dev/null | sed 's/.*engine \([0-9.]*\) ms.*/\1/' >> /tmp/ab_$L.txt
  done
done
cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so
for L in base "$@"; do sort -n /tmp/ab_$L.txt | awk -v f="$L" '{a[NR]=$1} END {printf "%-10s min %.2f  median %.2f  n=%d\n", f, a[1], (NR%2? a[(NR+1)/2] : (a[NR/2]+a[NR/2+1])/2), NR}'; done > $OUT/${TAG}_abhead.txt
