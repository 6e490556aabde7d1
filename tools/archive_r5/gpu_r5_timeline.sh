#!/bin/bash
# round 5: kernel TIMELINE of an n-frame 1080p batch on the segment engine at the library's defaults (launch groups side by side): tools/timeline.py over rocprofv3's trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
N=${1:-32}
TAG=${2:-r05_tl}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
cat > /tmp/bn.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import pngloss_amd as P
w, h, n = int(os.environ.get("TL_W", 1920)), int(os.environ.get("TL_H", 1080)), int(sys.argv[1])
ctx = P.HipContext()
base = [P.synth_rgba(w, h, 0, i) for i in range(4)]
ds = [torch.from_numpy(base[i % 4].copy()).cuda() for i in range(n)]
fs = [torch.zeros(h, dtype=torch.uint8, device="cuda") for i in range(n)]
torch.cuda.synchronize()
ctx.run([(d.data_ptr(), f.data_ptr(), w, h) for d, f in zip(ds, fs)], 19, 2)
print(n, ctx.engine_ms, ctx.engine_info(0))
PY
PNGLOSS_HIP_ENGINE=${TL_ENGINE:-seg} rocprofv3 --kernel-trace -d $OUT/${TAG}_prof -o trace --output-format csv -- python /tmp/bn.py $N > $OUT/${TAG}_prof.log 2>&1
{ echo "# rocprofv3 --kernel-trace, $N frames of ${TL_W:-1920}x${TL_H:-1080} s=19 b=2 in one batch, PNGLOSS_HIP_SEG_GROUPS=${PNGLOSS_HIP_SEG_GROUPS:-default}; tools/timeline.py"; grep "^$N " $OUT/${TAG}_prof.log; python tools/timeline.py $(find $OUT/${TAG}_prof -name "*kernel_trace.csv" | head -1) ${3:-} ${4:-}; } > $OUT/${TAG}_timeline_$N.txt 2>&1
rm -rf $OUT/${TAG}_prof
