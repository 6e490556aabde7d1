"""Coverage table of the segment-parallel engine (written to profiles/ by the builder): which (strength, bleed) pairs it takes and how
(exhaustive chain-state set / seeded enumeration), from the engine's own parameter builder run on the host (tests/c/seg_host.cpp), and --
on a GPU box -- measured throughput of an 8192-pixel-wide strip at selected pairs with the engine pinned, bytes compared with the
one-workgroup-per-image engine.   usage: gpu_seg_coverage.py [rows]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pngloss_amd as P  # noqa: E402
from tests import util as U  # noqa: E402

lib = U.seg_host_lib()
lib.seg_host_describe.argtypes = [C.c_uint, C.c_long, C.c_void_p]


def describe(s, b):
    o = (C.c_int32 * 8)()
    rc = lib.seg_host_describe(s, b, o)
    return rc, list(o)


print("# segment-parallel engine: (strength, bleed) coverage -- seg_build_params (pl_seg_core.h) for every strength 0..255")
print("# mode: E<n> = exhaustive enumeration of n chain states (<= 1024), S = seeded enumeration (256 seeds per channel, run-in 16 .. 32 pixels by the size of the carried terms)")
for b in (1, 2, 3, 4, 8, 16, 32767):
    runs, prev, start = [], None, 0
    for s in range(256):
        rc, o = describe(s, b)
        m = "unsupported" if rc else ("S" if o[1] else "E")
        if m != prev:
            if prev is not None:
                runs.append((start, s - 1, prev))
            prev, start = m, s
    runs.append((start, 255, prev))
    print("bleed %5d: " % b + ", ".join("s %d-%d %s" % r for r in runs))
print("# selected pairs: strength bleed mode states cmax tmax dmax")
for s, b in [(19, 2), (20, 1), (20, 2), (20, 8), (40, 1), (40, 2), (40, 8), (85, 1), (85, 2), (85, 8), (99, 2), (128, 1), (255, 1), (255, 2)]:
    rc, o = describe(s, b)
    print("  s=%3d b=%d: %s states=%d run-in=%d cmax=%d tmax=%d dmax=%d" % (s, b, "unsupported" if rc else ("seeded" if o[1] else "exhaustive"), o[2], o[3], o[4], o[5], o[6]))

try:
    import torch
    have_gpu = torch.cuda.is_available()
except Exception:
    have_gpu = False
if have_gpu:
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    print("# measured: 8192 x %d strip of the configs[4] frame (generator mode 0), engine pinned (PNGLOSS_HIP_ENGINE=seg), engine ms from the library's events" % rows)

    def run(engine, img, s, b):
        os.environ["PNGLOSS_HIP_ENGINE"] = engine
        h, w = img.shape[:2]
        ctx = P.HipContext()
        d = torch.from_numpy(img.copy()).cuda()
        f = torch.zeros(h, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.run([(d.data_ptr(), f.data_ptr(), w, h)], s, b)
        return d.cpu().numpy(), f.cpu().numpy(), ctx.engine_ms, ctx.engine_info(0)

    img = P.synth_rgba(8192, rows, 0, 0)
    run("seg", img, 19, 2)
    for s, b in [(0, 1), (20, 1), (20, 2), (20, 8), (40, 1), (40, 2), (40, 8), (85, 1), (85, 2), (85, 8), (60, 2), (99, 2), (128, 1), (255, 1), (255, 2)]:
        o1, f1, ms, info = run("seg", img, s, b)
        o2, f2, ms2, _ = run("wg", img, s, b)
        ok = np.array_equal(o1, o2) and np.array_equal(f1, f2)
        print("  s=%3d b=%d: %7.2f ms = %6.1f Mpx/s  engine=%s attempts=%d epochs=%d walked_segments=%d  (workgroup engine %.0f ms)  %s"
              % (s, b, ms, 8192 * rows / ms / 1e3, info["engine"], info["attempts"], info["restarts"], info["walked_segments"], ms2, "bytes+filters equal" if ok else "MISMATCH"), flush=True)
