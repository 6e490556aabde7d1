"""Check + timing aid (not a test): frames through the segment-parallel engine (pinned) at any strength / bleed, compared with the
one-workgroup-per-image engine's bytes and filters (both are exact; the latter is pinned to the reference elsewhere).
usage: gpu_seg_cover.py W,H,mode,s,b [W,H,mode,s,b ...]        (PNGLOSS_COVER_NOCHECK=1: timing only)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pngloss_amd as P  # noqa: E402
import torch  # noqa: E402


def run(engine, img, s, b):
    os.environ["PNGLOSS_HIP_ENGINE"] = engine
    h, w = img.shape[:2]
    ctx = P.HipContext()
    d = torch.from_numpy(img.copy()).cuda()
    f = torch.zeros(h, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.run([(d.data_ptr(), f.data_ptr(), w, h)], s, b)
    return d.cpu().numpy(), f.cpu().numpy(), ctx.engine_ms, ctx.engine_info(0)


bad = 0
for a in sys.argv[1:]:
    w, h, m, s, b = [int(v) for v in a.split(",")]
    img = P.synth_rgba(w, h, m, 0)
    run("seg", img, s, b)
    o1, f1, ms, info = run("seg", img, s, b)
    line = f"{w}x{h} mode {m} s={s} b={b}: seg {ms:.2f} ms = {w * h / ms / 1e3:.2f} Mpx/s  {info}"
    if not os.environ.get("PNGLOSS_COVER_NOCHECK"):
        o2, f2, ms2, info2 = run("wg", img, s, b)
        ok = np.array_equal(o1, o2) and np.array_equal(f1, f2)
        bad += not ok
        line += f"  | wg {ms2:.2f} ms  {'EQUAL' if ok else 'MISMATCH'}"
    print(line, flush=True)
sys.exit(1 if bad else 0)
