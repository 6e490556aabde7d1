import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PNGLOSS_HIP_DEBUG"] = "1"
import pngloss_amd as P
ctx = P.HipContext()
W, H = 640, 360
base = [P.synth_rgba(W, H, 0, f) for f in range(8)]
for n in (256, 512):
    dev = [torch.from_numpy(base[i % 8]).cuda() for i in range(n)]
    filt = [torch.zeros(H, dtype=torch.uint8, device="cuda") for _ in range(n)]
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.run([(d.data_ptr(), f.data_ptr(), W, H) for d, f in zip(dev, filt)], 19, 2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"== {n} frames: {dt:.3f} s engine {ctx.engine_ms:.1f} ms", flush=True)
