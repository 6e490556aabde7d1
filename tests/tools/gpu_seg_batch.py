"""Timing aid (not a test): n frames of WxH in ONE device-resident batch, once per row engine.  usage: gpu_seg_batch.py W H n [n ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import pngloss_amd as P
w, h, ns = %d, %d, %r
ctx = P.HipContext()
base = [P.synth_rgba(w, h, 0, i) for i in range(4)]
for n in ns:
    best = None
    for rep in range(2):
        ds = [torch.from_numpy(base[i %% 4].copy()).cuda() for i in range(n)]
        fs = [torch.zeros(h, dtype=torch.uint8, device="cuda") for i in range(n)]
        torch.cuda.synchronize()
        ctx.run([(d.data_ptr(), f.data_ptr(), w, h) for d, f in zip(ds, fs)], 19, 2)
        ms = ctx.engine_ms
        best = ms if best is None else min(best, ms)
    info = ctx.engine_info(0)
    print("n=%%3d  %%8.2f ms  %%8.1f Mpx/s  %%s attempts=%%s" %% (n, best, n * w * h / best / 1e3, info.get("engine"), info.get("attempts")))
'''
w, h = int(sys.argv[1]), int(sys.argv[2])
ns = [int(v) for v in sys.argv[3:]] or [1, 2, 4, 8, 16]
for eng in os.environ.get("SEG_BATCH_ENGINES", "seg,wg").split(","):
    print("--- PNGLOSS_HIP_ENGINE=%s  %dx%d" % (eng, w, h))
    r = subprocess.run([sys.executable, "-c", CODE % (ROOT, w, h, ns)], env=dict(os.environ, PNGLOSS_HIP_ENGINE=eng), capture_output=True, text=True, timeout=1500)
    print(r.stdout, r.stderr[-800:] if r.returncode else "")
