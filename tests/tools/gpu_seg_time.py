"""Timing aid (not a test): one frame through the device-resident batch API, engine milliseconds from the library's events.
usage: gpu_seg_time.py W H [mode] [strength] [bleed] [repeat]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pngloss_amd as P  # noqa: E402
import torch  # noqa: E402

w, h = int(sys.argv[1]), int(sys.argv[2])
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s = int(sys.argv[4]) if len(sys.argv) > 4 else 19
b = int(sys.argv[5]) if len(sys.argv) > 5 else 2
rep = int(sys.argv[6]) if len(sys.argv) > 6 else 2
img = P.synth_rgba(w, h, mode, 0)
ctx = P.HipContext()
for r in range(rep):
    d = torch.from_numpy(img.copy()).cuda()
    f = torch.zeros(h, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    res = ctx.run([(d.data_ptr(), f.data_ptr(), w, h)], s, b)
    ms = ctx.engine_ms
    out = d.cpu().numpy()
    print(f"{w}x{h} mode {mode} s={s} b={b}: engine {ms:.2f} ms = {w * h / ms / 1e3:.2f} Mpx/s  out={P.fnv1a64(out, P.SURVEY_FNV_BASIS):016x} filt={P.fnv1a64(f.cpu().numpy(), P.SURVEY_FNV_BASIS):016x}")
