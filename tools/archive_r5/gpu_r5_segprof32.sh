cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
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
print(n, ctx.engine_ms)
PY
PNGLOSS_HIP_ENGINE=seg PNGLOSS_HIP_SEGPROF=1 PNGLOSS_HIP_DEBUG=1 python /tmp/bn.py 32 2>&1 | grep -v amdgpu | head -14 | cut -c1-420 > gpurun_out/r05ak_segprof32.txt
