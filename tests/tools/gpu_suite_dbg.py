import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pngloss_amd as P
z = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden", "suite_inputs.npz"))
ctx = P.HipContext()
for k in sys.argv[1:]:
    img = np.ascontiguousarray(z[k]); h, w = img.shape[:2]
    d = torch.from_numpy(img.copy()).cuda(); f = torch.zeros(h, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.run([(d.data_ptr(), f.data_ptr(), w, h)], 19, 2)
    print(k, w, h, ctx.engine_ms, ctx.engine_info(0))
    fl = f.cpu().numpy(); print("filters hist", np.bincount(fl, minlength=5))
