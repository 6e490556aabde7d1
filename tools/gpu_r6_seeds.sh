#!/bin/bash
# round 6: the units of a batch from seeds (PNGLOSS_HIP_SEG_SEEDS=1, the default) against from every state (=0): rank shares of configs[3] on one GPU, interleaved
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r06_seeds}
NS=${NS:-"16 32 64 96 128"}
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}; python tests/tools/gpu_rank_share.py $NS  (PNGLOSS_HIP_ENGINE=seg; best of 2; wall = enqueue .. finish of the synchronous entry point)"
  for rep in 1 2; do
    for seeds in 0 1; do
      echo "## PNGLOSS_HIP_SEG_SEEDS=$seeds ${EXTRA:-}"
      env PNGLOSS_HIP_ENGINE=seg PNGLOSS_HIP_SEG_SEEDS=$seeds ${EXTRA:-} timeout 900 python tests/tools/gpu_rank_share.py $NS 2>&1 | grep -v amdgpu.ids
    done
  done
} > $OUT/${TAG}.txt 2>&1
