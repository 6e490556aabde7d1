#!/bin/bash
# round 6: the one campaign leg of the final run that printed no result line (seed 72, segments from seeds pinned), again with everything it says and its exit code
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  echo "## PNGLOSS_HIP_SEG_UNIT=0 PNGLOSS_HIP_SEG_SEEDS1=1 FUZZ_ENGINES=seg timeout 400 python tests/tools/gpu_fuzz.py 150 72"
  date +%s
  PNGLOSS_HIP_SEG_UNIT=0 PNGLOSS_HIP_SEG_SEEDS1=1 FUZZ_ENGINES=seg timeout 400 python tests/tools/gpu_fuzz.py 150 72 2>&1 | grep -v amdgpu.ids | tail -20
  echo "rc=${PIPESTATUS[0]}"
  date +%s
  dmesg 2>/dev/null | tail -5
} > $OUT/r06_fuzz72.txt 2>&1
