"""How does batch throughput scale with the number of images (= workgroups) on one MI355X?  1920x1080 frames (BASELINE.json
configs[3] shape), band-leader chains and (PNGLOSS_HIP_ENGINE=legacy) the round-1 chains.
usage: python tools/gpu_batch_scale.py [n1,n2,...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pngloss_amd as P
ctx = P.HipContext()
W, H = 1920, 1080
ns = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 256, 1024, 4096]
base = [P.synth_rgba(W, H, 0, f) for f in range(8)]
tag = os.environ.get("PNGLOSS_HIP_ENGINE", "band-leader")
for n in ns:
    dev = [torch.from_numpy(base[i % 8]).cuda() for i in range(n)]
    filt = [torch.zeros(H, dtype=torch.uint8, device="cuda") for _ in range(n)]
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.run([(d.data_ptr(), f.data_ptr(), W, H) for d, f in zip(dev, filt)], 19, 2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"[{tag}] {n:5d} frames {W}x{H}: {dt:.3f} s  {n*W*H/dt/1e6:8.1f} Mpx/s  engine {ctx.engine_ms:.1f} ms", flush=True)
    del dev, filt
    torch.cuda.empty_cache()
