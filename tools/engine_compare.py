"""Developer tool (GPU box): band-leader engine vs the round-1 chains (PNGLOSS_HIP_ENGINE=legacy, separate process) on the suite
images and on synthetic frames at several strengths -- is the new engine ever slower?   usage: python tools/engine_compare.py [legacy]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pngloss_amd as P
tag = os.environ.get("PNGLOSS_HIP_ENGINE", "band-leader")
cases = []
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "suite_inputs.npz"))
for k in sorted(z.files):
    cases.append(("suite/" + k, z[k], 19, 2))
for s in (5, 19, 40, 85, 127):
    cases.append((f"synth 2048x2048 mode 0 s={s}", P.synth_rgba(2048, 2048, 0, 0), s, 2))
for m in (1, 3, 5):
    cases.append((f"synth 2048x2048 mode {m} s=19", P.synth_rgba(2048, 2048, m, 0), 19, 2))
P.optimize_with_rows(P.synth_rgba(64, 8, 0, 0), 19, 2)
for name, img, s, b in cases:
    img = np.ascontiguousarray(img)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); P.optimize_with_rows(img, s, b); best = min(best, time.perf_counter() - t)
    h, w = img.shape[:2]
    print(f"[{tag}] {name:34s} {w}x{h}: {best*1e3:8.1f} ms  {w*h/best/1e6:6.2f} Mpx/s", flush=True)
