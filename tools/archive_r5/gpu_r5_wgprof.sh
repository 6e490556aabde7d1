#!/bin/bash
# Developer tool (GPU box): the workgroup-per-image engine built with -DPL_LEAD_PROF=1 (tools/build_wg_variant.sh prof "" "-DPL_LEAD_PROF=1") on single 1080p frames 0 and 7:
# the per-phase cycle counters of the chain waves (vector, fast groups, exact redo, rescan, table build) beside the repaired-pixel counts
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp tools/ablate_build/libpngloss_hip_prof.so pngloss_amd/csrc/libpngloss_hip.so
cat > /tmp/w.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import pngloss_amd as P
w, h = 1920, 1080
ctx = P.HipContext()
for i in (0, 7):
    a = P.synth_rgba(w, h, 0, i)
    d = torch.from_numpy(a.copy()).cuda(); f = torch.zeros(h, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.run([(d.data_ptr(), f.data_ptr(), w, h)], 19, 2)
    print("frame", i, "engine ms %.1f" % ctx.engine_ms, flush=True)
PY
PNGLOSS_HIP_ENGINE=wg PNGLOSS_HIP_DEBUG=1 python /tmp/w.py 2>&1 | grep -E "image 0:|engine ms|kcycles|wave" | cut -c1-300 > gpurun_out/r05w_wgprof.txt
