#!/bin/bash
# round 6 experiment: ONE image segment by segment from seeds (PNGLOSS_HIP_SEG_SEEDS1=1) with 2 / 4 / 8 pairs a workgroup against the enumeration from every state (headline frame strip 4096x2048)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  for rep in 1 2; do
  echo "## from every state (shipped for one image)"; python tests/tools/gpu_seg_time.py 4096 2048 0 19 2 3 2>&1 | grep Mpx | tail -1
  for lib in ../../tools/ablate_build/libpngloss_hip_s1nc2.so ../../tools/ablate_build/libpngloss_hip_s1nc4.so libpngloss_hip.so; do
    echo "## PNGLOSS_HIP_SEG_SEEDS1=1 LIB=$lib"; PNGLOSS_HIP_LIBNAME=$lib PNGLOSS_HIP_SEG_SEEDS1=1 PNGLOSS_HIP_DEBUG=1 python tests/tools/gpu_seg_time.py 4096 2048 0 19 2 3 2>&1 | grep "Mpx\|attempts" | tail -2 | cut -c1-200
  done; done
} > $OUT/r06_head_seeds.txt 2>&1
