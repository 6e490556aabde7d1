#!/bin/bash
# round 6: unit length with seeds (build variants)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}: SEG_UNIT 3 (shipped) / 2 / 4 with seeds, units pinned on"
  for lib in libpngloss_hip.so ../../tools/ablate_build/libpngloss_hip_unit2.so ../../tools/ablate_build/libpngloss_hip_unit4.so libpngloss_hip.so; do
    echo "## LIB=$lib"
    PNGLOSS_HIP_LIBNAME=$lib PNGLOSS_HIP_SEG_UNIT=1 PNGLOSS_HIP_ENGINE=seg timeout 900 python tests/tools/gpu_rank_share.py 16 24 32 48 64 128 2>&1 | grep -v amdgpu.ids
  done
} > $OUT/r06_units.txt 2>&1
