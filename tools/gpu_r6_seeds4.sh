#!/bin/bash
# round 6: per-segment seeds with 8 pairs a workgroup (one turn) against 16
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  for lib in libpngloss_hip.so ../../tools/ablate_build/libpngloss_hip_s1nc8.so; do
    echo "## LIB=$lib PNGLOSS_HIP_SEG_UNIT=0 (per segment from seeds at every size)"
    PNGLOSS_HIP_LIBNAME=$lib PNGLOSS_HIP_SEG_UNIT=0 PNGLOSS_HIP_ENGINE=seg timeout 900 python tests/tools/gpu_rank_share.py 2 4 6 8 11 12 16 20 24 32 2>&1 | grep -v amdgpu.ids
    echo "## LIB=$lib suite + small"
    PNGLOSS_HIP_LIBNAME=$lib timeout 600 python tests/tools/gpu_small_batches.py 2 2>&1 | grep -v amdgpu.ids
  done
} > $OUT/r06_seeds_d.txt 2>&1
