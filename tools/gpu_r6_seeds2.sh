#!/bin/bash
# round 6: seeds -- pairs per workgroup (build variants), run-in length, kernel traces
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}: seeds variants, rank shares 32 / 64 / 128 (best of 2)"
  for lib in libpngloss_hip.so ../../tools/ablate_build/libpngloss_hip_unc8.so ../../tools/ablate_build/libpngloss_hip_unc12.so ../../tools/ablate_build/libpngloss_hip_unc20.so libpngloss_hip.so; do
    echo "## LIB=$lib"
    PNGLOSS_HIP_LIBNAME=$lib PNGLOSS_HIP_ENGINE=seg timeout 600 python tests/tools/gpu_rank_share.py 32 64 128 2>&1 | grep -v amdgpu.ids
  done
  for kin in 4 6 10 12 16; do
    echo "## PNGLOSS_HIP_SEED_KIN=$kin"
    PNGLOSS_HIP_SEED_KIN=$kin PNGLOSS_HIP_ENGINE=seg timeout 600 python tests/tools/gpu_rank_share.py 32 64 128 2>&1 | grep -v amdgpu.ids
  done
} > $OUT/r06_seeds_b.txt 2>&1
PNGLOSS_HIP_SEG_GROUPS=1 bash tools/gpu_r6_prof.sh 11 r06_g1_seeds
PNGLOSS_HIP_SEG_GROUPS=1 PNGLOSS_HIP_SEG_SEEDS=0 bash tools/gpu_r6_prof.sh 11 r06_g1_exh
bash tools/gpu_r6_prof.sh 64 r06_64_seeds
PNGLOSS_HIP_SEG_SEEDS=0 bash tools/gpu_r6_prof.sh 64 r06_64_exh
