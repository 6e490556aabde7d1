"""Developer tool (GPU box): time one synthetic frame through the row engine and print the engine's diagnostics.
usage: PNGLOSS_HIP_DEBUG=1 python tools/lead_time.py W H [mode] [s] [b]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pngloss_amd import lib, synth
W, H = int(sys.argv[1]), int(sys.argv[2])
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s = int(sys.argv[4]) if len(sys.argv) > 4 else 19
b = int(sys.argv[5]) if len(sys.argv) > 5 else 2
img = synth.synth_rgba(W, H, mode, 0)
lib.optimize_with_rows(synth.synth_rgba(64, 8, 0, 0), s, b)
t = time.time()
out, f = lib.optimize_with_rows(img, s, b)
dt = time.time() - t
print(f"{W}x{H} mode {mode} s={s} b={b}: host call {dt*1e3:.1f} ms = {W*H/dt/1e6:.2f} Mpx/s  out digest {synth.fnv1a64(out, synth.SURVEY_FNV_BASIS):016x}")
