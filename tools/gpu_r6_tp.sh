#!/bin/bash
# round 6 experiment: two control workgroups per candidate for batches composed in units (-DSEG_TPARTS_BATCH=2) against one
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  for lib in libpngloss_hip.so ../../tools/ablate_build/libpngloss_hip_tp2.so libpngloss_hip.so ../../tools/ablate_build/libpngloss_hip_tp2.so; do
    echo "## LIB=$lib"; PNGLOSS_HIP_LIBNAME=$lib PNGLOSS_HIP_ENGINE=seg timeout 600 python tests/tools/gpu_rank_share.py 24 32 48 64 128 2>&1 | grep -v amdgpu.ids
  done
} > $OUT/r06_tparts.txt 2>&1
