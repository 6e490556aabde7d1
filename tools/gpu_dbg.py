import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PNGLOSS_HIP_DEBUG"] = "1"
import pngloss_amd as P
for (w, h, m, s) in [(1920, 1080, 0, 19), (1920, 1080, 1, 19), (1920, 1080, 2, 19), (1920, 1080, 4, 19), (1920,1080,0,7), (1920,1080,0,40)]:
    img = P.synth_rgba(w, h, m, 0)
    t = time.time(); P.optimize_with_rows(img, s, 2); dt = time.time() - t
    print(f"mode {m} s{s}: {w*h/dt/1e6:.2f} Mpx/s host-call", flush=True)
