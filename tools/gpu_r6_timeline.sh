#!/bin/bash
# round 6: kernel TIMELINE (tools/timeline.py over rocprofv3 --kernel-trace) of the first n frames of configs[3] as one batch, as bench.py's batch legs run them
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${TAG:-r06_tl}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
for N in "$@"; do
  PNGLOSS_HIP_ENGINE=seg SHARE_REPS=1 rocprofv3 --kernel-trace -d $OUT/${TAG}_prof -o trace --output-format csv -- python tests/tools/gpu_rank_share.py $N > $OUT/${TAG}_prof.log 2>&1
  { echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
    echo "# rocprofv3 --kernel-trace -- python tests/tools/gpu_rank_share.py $N (PNGLOSS_HIP_ENGINE=seg, three launch groups opted in); tools/timeline.py"; grep "^n=" $OUT/${TAG}_prof.log
    python tools/timeline.py $(find $OUT/${TAG}_prof -name "*kernel_trace.csv" | head -1); } > $OUT/${TAG}_timeline_$N.txt 2>&1
  rm -rf $OUT/${TAG}_prof
done
