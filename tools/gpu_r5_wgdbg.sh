cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > /tmp/w.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import pngloss_amd as P
w, h = 1920, 1080
ctx = P.HipContext()
img = P.synth_rgba(w, h, 0, 0)
d = torch.from_numpy(img.copy()).cuda(); f = torch.zeros(h, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.run([(d.data_ptr(), f.data_ptr(), w, h)], 19, 2)
print(ctx.engine_ms, ctx.engine_info(0))
PY
PNGLOSS_HIP_ENGINE=wg PNGLOSS_HIP_DEBUG=1 python /tmp/w.py > gpurun_out/r05l_wgdbg.txt 2>&1
