"""Developer tool (GPU box): band-leader row engine vs the CPU oracle on a set of synthetic cases; prints the first
mismatching row/pixel.  usage: python tools/lead_check.py [quick|full]"""
import os, sys, time
os.environ.setdefault("PNGLOSS_HIP_ENGINE", "lead")   # this tool is about the band-leader chains: no adaptive fallback to the round-1 chains
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pngloss_amd import lib, synth
from tests import util

def check(name, img, s=19, b=2, filters=True):
    t = time.time()
    out, f = lib.optimize_with_rows(img, s, b, want_filters=filters)
    dt = time.time() - t
    ro, rf = util.run_port(img, s, b, filters)
    ok = np.array_equal(out, ro) and (not filters or np.array_equal(f, rf))
    msg = f"{name:34s} s={s:3d} b={b:5d} {'ok ' if ok else 'MISMATCH'} {dt*1e3:8.1f} ms"
    if not ok:
        rows = np.where((out != ro).any(axis=(1, 2)))[0]
        if len(rows):
            y = rows[0]; xs = np.where((out[y] != ro[y]).any(axis=1))[0]
            msg += f" first diff row {y} x {xs[0]} got {out[y, xs[0]]} want {ro[y, xs[0]]} ({len(rows)} rows differ)"
        if filters and not np.array_equal(f, rf):
            fy = np.where(f != rf)[0][0]
            msg += f" | first filter diff row {fy}: got {f[fy]:#x} want {rf[fy]:#x}"
    print(msg, flush=True)
    return ok

def main():
    full = len(sys.argv) > 1 and sys.argv[1] == "full"
    ok = True
    for (w, h) in [(1, 1), (5, 1), (1, 7), (2, 3), (17, 5), (64, 48), (63, 9), (65, 9), (130, 20), (200, 64)]:
        for mode in range(6):
            ok &= check(f"synth {w}x{h} mode {mode}", synth.synth_rgba(w, h, mode, 0))
    for s in [0, 1, 2, 3, 7, 15, 16, 31, 40, 47, 48, 63, 85, 100, 127, 128, 200, 255]:
        ok &= check("synth 96x40 mode 0", synth.synth_rgba(96, 40, 0, 1), s, 2)
        ok &= check("synth 96x40 mode 5", synth.synth_rgba(96, 40, 5, 1), s, 2)
        ok &= check("synth 96x40 mode 1", synth.synth_rgba(96, 40, 1, 1), s, 1)
    for b in [1, 3, 8, 100, 32767]:
        ok &= check("synth 160x40 mode 0", synth.synth_rgba(160, 40, 0, 2), 19, b)
    ok &= check("synth 96x64 null filters", synth.synth_rgba(96, 64, 0, 3), 19, 2, filters=False)
    if full:
        for mode in range(6):
            ok &= check(f"synth 512x512 mode {mode}", synth.synth_rgba(512, 512, mode, 0))
        ok &= check("synth 1920x270 mode 0", synth.synth_rgba(1920, 270, 0, 0))
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1

if __name__ == "__main__":
    sys.exit(main())
