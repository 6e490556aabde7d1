#!/bin/bash
# round 6: per-segment enumeration from seeds (seg_k_enum_unit<1>) against the round-5 paths: small batches, mid-size rank shares (units pinned off / on), the suite batch
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  echo "## A. library defaults (per segment from seeds up to 680 segments, units from seeds beyond)"
  PNGLOSS_HIP_ENGINE=seg timeout 600 python tests/tools/gpu_rank_share.py 2 4 8 11 12 16 24 32 48 64 2>&1 | grep -v amdgpu.ids
  echo "## B. PNGLOSS_HIP_SEG_SEEDS1=0 (round-5 per-segment enumeration for small batches)"
  PNGLOSS_HIP_SEG_SEEDS1=0 PNGLOSS_HIP_ENGINE=seg timeout 600 python tests/tools/gpu_rank_share.py 2 4 8 11 2>&1 | grep -v amdgpu.ids
  echo "## C. PNGLOSS_HIP_SEG_UNIT=0 (per segment at every size): from seeds"
  PNGLOSS_HIP_SEG_UNIT=0 PNGLOSS_HIP_ENGINE=seg timeout 900 python tests/tools/gpu_rank_share.py 12 16 24 32 48 64 2>&1 | grep -v amdgpu.ids
  echo "## D. PNGLOSS_HIP_SEG_UNIT=1 (units at every size): from seeds"
  PNGLOSS_HIP_SEG_UNIT=1 PNGLOSS_HIP_ENGINE=seg timeout 900 python tests/tools/gpu_rank_share.py 4 8 11 2>&1 | grep -v amdgpu.ids
  echo "## E. suite + small mixed batches: defaults, then PNGLOSS_HIP_SEG_SEEDS1=0"
  timeout 600 python tests/tools/gpu_small_batches.py 2 2>&1 | grep -v amdgpu.ids
  PNGLOSS_HIP_SEG_SEEDS1=0 timeout 600 python tests/tools/gpu_small_batches.py 2 2>&1 | grep -v amdgpu.ids
} > $OUT/r06_seeds_c.txt 2>&1
