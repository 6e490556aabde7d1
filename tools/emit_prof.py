import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pngloss_amd as P
ctx = P.HipContext()
for mode in (0, 2, 4):
    img = P.synth_rgba(4096, 256, mode, 0)
    big = np.tile(img, (8, 1, 1))      # 4096 x 2048: the emit kernels see 32 MiB, the engine dominates the wall time anyway
    ctx.run_host_emit([big], 0, 2)
print("done")
