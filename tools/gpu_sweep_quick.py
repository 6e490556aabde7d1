import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PNGLOSS_HIP_DEBUG"] = "1"
import pngloss_amd as P
for s in (0, 7, 19, 20, 40, 63, 85, 128, 255):
    img = P.synth_rgba(1920, 1080, 0, 0)
    t = time.time(); P.optimize_with_rows(img, s, 2); dt = time.time() - t
    print(f"s={s}: {1920*1080/dt/1e6:.2f} Mpx/s host-call", flush=True)
