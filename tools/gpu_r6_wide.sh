#!/bin/bash
# round 6: seeds for state sets of several chunks (one image, segment by segment): 8192x2048 strips of the configs[4] frame, PNGLOSS_HIP_SEG_SEEDS1=0 (round 5) against the default
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  for sb in "20 1" "40 2" "85 8" "40 8" "19 2"; do
    for seeds in 0 default; do
      echo "## s b = $sb   PNGLOSS_HIP_SEG_SEEDS1=$seeds"
      if [ $seeds = 0 ]; then PNGLOSS_HIP_SEG_SEEDS1=0 PNGLOSS_HIP_DEBUG=1 timeout 600 python tests/tools/gpu_seg_time.py 8192 2048 0 $sb 2 2>&1 | grep -v amdgpu.ids | grep "Mpx\|attempts" | cut -c1-250
      else PNGLOSS_HIP_DEBUG=1 timeout 600 python tests/tools/gpu_seg_time.py 8192 2048 0 $sb 2 2>&1 | grep -v amdgpu.ids | grep "Mpx\|attempts" | cut -c1-250; fi
    done
  done
} > $OUT/r06_wide.txt 2>&1
