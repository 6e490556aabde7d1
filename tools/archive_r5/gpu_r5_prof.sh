#!/bin/bash
# round 5: kernel trace of an n-frame 1080p batch on the segment engine (units on/off by env)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
N=${1:-32}
TAG=${2:-r05_unit}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
cat > /tmp/bn.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import pngloss_amd as P
w, h, n = 1920, 1080, int(sys.argv[1])
ctx = P.HipContext()
base = [P.synth_rgba(w, h, 0, i) for i in range(4)]
ds = [torch.from_numpy(base[i % 4].copy()).cuda() for i in range(n)]
fs = [torch.zeros(h, dtype=torch.uint8, device="cuda") for i in range(n)]
torch.cuda.synchronize()
ctx.run([(d.data_ptr(), f.data_ptr(), w, h) for d, f in zip(ds, fs)], 19, 2)
print(n, ctx.engine_ms, ctx.engine_info(0))
PY
PNGLOSS_HIP_ENGINE=seg rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o trace --output-format csv -- python /tmp/bn.py $N > $OUT/${TAG}_prof.log 2>&1
{ echo "# PNGLOSS_HIP_ENGINE=seg PNGLOSS_HIP_SEG_UNIT=${PNGLOSS_HIP_SEG_UNIT:-default} rocprofv3 --kernel-trace --stats -- $N frames of 1920x1080 s=19 b=2 in one batch"; cat $(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1) | cut -c1-200 | head -8; } > $OUT/${TAG}_kernel_stats_seg$N.txt
rm -rf $OUT/${TAG}_prof
