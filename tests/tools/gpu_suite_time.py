"""Timing aid (not a test): the eleven suite images (tests/golden/suite_inputs.npz) one at a time through the device-resident API,
once per row engine; engine milliseconds from the library's events, attempts / epochs from pngloss_hip_last_engine_info.
usage: gpu_suite_time.py [strength] [bleed]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CODE = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import pngloss_amd as P
s, b = %d, %d
z = np.load(os.path.join(%r, "tests", "golden", "suite_inputs.npz"))
ctx = P.HipContext()
for k in z.files:
    img = np.ascontiguousarray(z[k]); h, w = img.shape[:2]
    best = None
    for rep in range(2):
        d = torch.from_numpy(img.copy()).cuda(); f = torch.zeros(h, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.run([(d.data_ptr(), f.data_ptr(), w, h)], s, b)
        ms = ctx.engine_ms
        best = ms if best is None else min(best, ms)
    info = ctx.engine_info(0)
    print("%%-28s %%5dx%%-5d %%8.2f ms %%7.2f Mpx/s  %%s attempts=%%s epochs=%%s serial=%%s digest=%%016x" %% (k, w, h, best, w * h / best / 1e3, info.get("engine"), info.get("attempts"), info.get("restarts"), info.get("serial_rows"), P.fnv1a64(d.cpu().numpy(), P.SURVEY_FNV_BASIS)))
'''
s = int(sys.argv[1]) if len(sys.argv) > 1 else 19
b = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for eng in ("seg", "wg"):
    print("--- PNGLOSS_HIP_ENGINE=%s  s=%d b=%d" % (eng, s, b))
    env = dict(os.environ, PNGLOSS_HIP_ENGINE=eng)
    r = subprocess.run([sys.executable, "-c", CODE % (ROOT, s, b, ROOT)], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout, r.stderr[-800:] if r.returncode else "")
