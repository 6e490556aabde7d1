"""Developer tool (GPU box): force one candidate filter to win every row, on the GPU and in the oracle, and compare.
usage: python tools/lead_force.py W H mode [s]"""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 5:
    from pngloss_amd import lib, synth
    from tests import util
    W, H, mode, s, f = [int(v) for v in sys.argv[1:6]]
    img = synth.synth_rgba(W, H, mode, 0)
    out, fl = lib.optimize_with_rows(img, s, 2)
    util.port().port_set_force_filter(f)
    ro, rf = util.run_port(img, s, 2, True)
    ok = np.array_equal(out, ro)
    msg = f"filter {f}: {'ok' if ok else 'MISMATCH'}"
    if not ok:
        rows = np.where((out != ro).any(axis=(1, 2)))[0]
        y = rows[0]; xs = np.where((out[y] != ro[y]).any(axis=1))[0]
        msg += f" first diff row {y} x {xs[:8]} got {out[y, xs[0]]} want {ro[y, xs[0]]} ({len(rows)} rows differ); prev px got {out[y, max(xs[0]-1,0)]} in {img[y, xs[0]]}"
    print(msg)
else:
    W, H, mode = sys.argv[1:4]
    s = sys.argv[4] if len(sys.argv) > 4 else "19"
    for f in range(5):
        env = dict(os.environ, PNGLOSS_HIP_FORCE_FILTER=str(f))
        subprocess.run([sys.executable, __file__, W, H, mode, s, str(f)], env=env)
