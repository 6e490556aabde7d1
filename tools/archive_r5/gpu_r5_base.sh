#!/bin/bash
# round 5 baseline: where does a rank's share of configs[3] (32 / 64 / 128 frames of 1920x1080) stand on ONE GPU, per engine, and which kernels take the time
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
python tests/tools/gpu_seg_batch.py 1920 1080 8 16 32 64 128 > $OUT/r05_base_batch.txt 2>&1
cat > /tmp/b32.py <<'PY'
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
PNGLOSS_HIP_ENGINE=seg rocprofv3 --kernel-trace --stats -d $OUT/r05_base_prof -o trace --output-format csv -- python /tmp/b32.py 32 > $OUT/r05_base_prof.log 2>&1
{ echo "# PNGLOSS_HIP_ENGINE=seg rocprofv3 --kernel-trace --stats -- 32 frames of 1920x1080 s=19 b=2 in one batch (round-4 code)"; cat $(find $OUT/r05_base_prof -name "*kernel_stats.csv" | head -1); } > $OUT/r05_base_kernel_stats_seg32.txt
rm -rf $OUT/r05_base_prof
