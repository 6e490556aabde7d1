#!/bin/bash
# round 6: the crossover between the engines around 150 frames of 1080p, then the GPU suite and the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  for eng in seg wg; do echo "## PNGLOSS_HIP_ENGINE=$eng"; PNGLOSS_HIP_ENGINE=$eng SHARE_REPS=1 timeout 900 python tests/tools/gpu_rank_share.py 128 144 160 2>&1 | grep -v amdgpu.ids; done
  echo "## the library's choice"; SHARE_REPS=1 timeout 900 python tests/tools/gpu_rank_share.py 6 16 32 64 128 144 160 2>&1 | grep -v amdgpu.ids
} > $OUT/r06_crossover.txt 2>&1
HEAD=${HEAD:-unknown} bash tools/gpu_round6.sh r06b "tests bench"
