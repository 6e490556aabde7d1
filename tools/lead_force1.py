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
for y in range(min(H, 3)):
    print("row", y, "in  a:", img[y, :, 3]); print("      got a:", out[y, :, 3]); print("     want a:", ro[y, :, 3])
    print("      in  g:", img[y, :, 1]); print("      got g:", out[y, :, 1]); print("     want g:", ro[y, :, 1])
