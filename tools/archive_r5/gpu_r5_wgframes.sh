#!/bin/bash
# Developer tool (GPU box): the workgroup-per-image engine on single 1080p frames of the generator (frame index 0 .. 7): engine ms and the engine's own statistics
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > /tmp/w.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import pngloss_amd as P
w, h = 1920, 1080
ctx = P.HipContext()
for i in range(8):
    a = P.synth_rgba(w, h, 0, i)
    d = torch.from_numpy(a.copy()).cuda(); f = torch.zeros(h, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.run([(d.data_ptr(), f.data_ptr(), w, h)], 19, 2)
    print("frame", i, "engine ms %.1f" % ctx.engine_ms, flush=True)
PY
PNGLOSS_HIP_ENGINE=wg PNGLOSS_HIP_DEBUG=1 python /tmp/w.py 2>&1 | grep -E "image 0: chain|image 0: band|light pixels|engine ms" | cut -c1-250 > gpurun_out/r05v_wgframes.txt
