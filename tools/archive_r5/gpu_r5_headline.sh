#!/bin/bash
# kernel trace of the headline (one 4096x4096 frame), for comparing per-kernel averages with profiles/r04_kernel_trace_stats.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05_head}
cd /tmp && export TMPDIR=/tmp
cd $R
for i in 1 2 3; do python tests/tools/gpu_seg_time.py 4096 4096 0 19 2 2 | tail -1; done > $OUT/${TAG}_time.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o trace --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-sweep > $OUT/${TAG}_prof.log 2>&1
cat $(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1) | cut -c1-200 | head -8 > $OUT/${TAG}_kernel_stats.txt
rm -rf $OUT/${TAG}_prof
