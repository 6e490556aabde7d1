"""Developer tool (GPU box): host-pointer batch call vs device-resident batch for a window of 256 synthetic 1280x720 files
(the shape of the command line's windows).  usage: python tools/host_seam_bench.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pngloss_amd as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = 1280, 720
frames = [P.synth_rgba(W, H, 0, i) for i in range(n)]
ctx = P.HipContext()
ctx.run_host(frames[:4], 19, 2)                     # warm-up: arena, pinned staging, code objects
for rep in range(3):
    work = [f.copy() for f in frames]                 # the call optimises in place, like the C entry point it wraps
    t = time.perf_counter(); ctx.run_host(work, 19, 2, inplace=True); dt = time.perf_counter() - t
    print(f"host-pointer batch of {n} x {W}x{H}: {dt:.3f} s = {n*W*H/dt/1e6:.1f} Mpx/s (engine {ctx.engine_ms:.1f} ms)")
dev = [torch.from_numpy(f).cuda() for f in frames]
filt = [torch.zeros(H, dtype=torch.uint8, device="cuda") for _ in frames]
torch.cuda.synchronize()
t = time.perf_counter(); ctx.run([(d.data_ptr(), f.data_ptr(), W, H) for d, f in zip(dev, filt)], 19, 2); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"device-resident batch of {n} x {W}x{H}: {dt:.3f} s = {n*W*H/dt/1e6:.1f} Mpx/s (engine {ctx.engine_ms:.1f} ms)")
