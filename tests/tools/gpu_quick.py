"""Quick GPU bring-up check (not a test): HIP path vs the CPU oracle on seeded inputs + first timings.
Run on the GPU box:  python tests/tools/gpu_quick.py [--big]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pngloss_amd as P  # noqa: E402

port = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "libpngloss_port.so"))
port.port_optimize_with_rows.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_void_p, C.c_bool, C.c_uint8, C.c_long]
port.port_optimize_with_rows.restype = C.c_int


def run_port(img, s, b, filters=True):
    h, w, _ = img.shape
    out = img.copy()
    f = np.zeros(h, np.uint8)
    rows = (C.c_void_p * h)(*[out.ctypes.data + y * w * 4 for y in range(h)])
    assert port.port_optimize_with_rows(rows, w, h, f.ctypes.data if filters else None, False, s, b) == 0
    return out, (f if filters else None)


def main():
    bad = 0
    n = 0
    cases = []
    for m in range(6):
        for (s, b) in [(19, 2), (0, 2), (20, 1), (40, 2), (85, 8), (255, 1), (7, 3)]:
            cases.append((64, 48, m, s, b, True))
    for wh in [(1, 1), (2, 3), (5, 1), (1, 7), (17, 5), (130, 9), (65, 3), (64, 2), (63, 2)]:
        for m in (1, 0, 3, 4, 5):
            cases.append((wh[0], wh[1], m, 19, 2, True))
            cases.append((wh[0], wh[1], m, 19, 2, False))
    for m in range(6):
        cases.append((96, 64, m, 19, 2, False))
    t0 = time.time()
    for (w, h, m, s, b, filt) in cases:
        img = P.synth_rgba(w, h, m, 0)
        o1, f1 = run_port(img, s, b, filt)
        o2, f2 = P.optimize_with_rows(img, s, b, want_filters=filt)
        n += 1
        ok = np.array_equal(o1, o2) and (not filt or np.array_equal(f1, f2))
        if not ok:
            bad += 1
            if bad <= 12:
                dy = np.nonzero((o1 != o2).any(axis=(1, 2)))[0]
                print("MISMATCH", (w, h, m, s, b, filt), "px diff", int((o1 != o2).sum()), "first bad row", dy[:3],
                      "filters", None if not filt else (f1[:6], f2[:6]))
    print(f"parity: {n} cases, {bad} bad, {time.time() - t0:.1f}s")
    sizes = [(512, 512), (1920, 1080)] + ([(4096, 4096)] if "--big" in sys.argv else [])
    for (w, h) in sizes:
        img = P.synth_rgba(w, h, 0, 0)
        t = time.time()
        o2, f2 = P.optimize_with_rows(img, 19, 2)
        dt = time.time() - t
        print(f"{w}x{h} s19 b2: host-call {dt:.3f}s = {w * h / dt / 1e6:.2f} Mpx/s  out={P.fnv1a64(o2, P.SURVEY_FNV_BASIS):016x} filt={P.fnv1a64(f2, P.SURVEY_FNV_BASIS):016x}")
        if w * h <= 1920 * 1080:
            t = time.time()
            o1, f1 = run_port(img, 19, 2)
            print(f"   port {time.time() - t:.2f}s  equal={np.array_equal(o1, o2)} {np.array_equal(f1, f2)}")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
