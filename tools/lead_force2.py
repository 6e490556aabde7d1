import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pngloss_amd import lib, synth
from tests import util
W, H, mode, s, f = [int(v) for v in sys.argv[1:6]]
img = synth.synth_rgba(W, H, mode, 0)
out, fl = lib.optimize_with_rows(img, s, 2)
util.port().port_set_force_filter(f)
ro, rf = util.run_port(img, s, 2, True)
np.set_printoptions(linewidth=250)
bad = np.argwhere((out != ro).any(axis=2))
print("W", W, "H", H, "mode", mode, "f", f, "mismatching pixels", len(bad), "first", bad[:12].tolist())
if len(bad):
    y, x = bad[0]
    x0 = max(0, x - 4)
    print(" in  ", img[y, x0:x0 + 12].tolist()); print(" got ", out[y, x0:x0 + 12].tolist()); print(" want", ro[y, x0:x0 + 12].tolist())
