#!/bin/bash
# Developer tool (GPU box): the workgroup-per-image engine's own statistics (PNGLOSS_HIP_DEBUG=1: chain cycles per wave, repaired pixels) for 1, 2, 4 and 16 frames of 1080p in one batch:
# does a frame take more CYCLES next to others, or the same cycles at a lower clock?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > /tmp/w.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import pngloss_amd as P
w, h, n = 1920, 1080, int(sys.argv[1])
ctx = P.HipContext()
imgs = [P.synth_rgba(w, h, 0, 0) for i in range(n)]
ds = [torch.from_numpy(a.copy()).cuda() for a in imgs]; fs = [torch.zeros(h, dtype=torch.uint8, device="cuda") for _ in imgs]
torch.cuda.synchronize()
ctx.run([(d.data_ptr(), f.data_ptr(), w, h) for d, f in zip(ds, fs)], 19, 2)
print("n =", n, "engine ms", ctx.engine_ms)
PY
for N in 1 2 4 16; do PNGLOSS_HIP_ENGINE=wg PNGLOSS_HIP_DEBUG=1 python /tmp/w.py $N 2>&1 | grep -E "image 0:|image 1:|image 3:|engine ms|SIMD" | cut -c1-260; done > gpurun_out/r05s_wgdbg.txt 2>&1
