"""Timing aid: pngloss_hip_png_decode_batch_host on one large random scanline stream per pixel format (whole call: upload + unfilter + expand +
download, from pageable memory), and the kernel alone when run under rocprofv3.  usage: read_formats.py [W H]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pngloss_amd as P
from pngloss_amd import lib as L
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
lib = P.hip_lib()
lib.pngloss_hip_png_decode_batch_host.restype = C.c_int
lib.pngloss_hip_png_decode_batch_host.argtypes = [C.c_void_p, C.POINTER(L.PngSource), C.c_size_t]
ctx = P.HipContext()
rng = np.random.default_rng(1)
for ctype, depth in [(6, 8), (2, 8), (0, 8), (4, 8), (6, 16), (2, 16), (3, 8), (0, 1)]:
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    rowbytes = (w * channels * depth + 7) // 8
    rows = rng.integers(0, 256, (h, 1 + rowbytes), dtype=np.uint8)
    rows[:, 0] = rng.choice([0, 1, 2, 3, 4, 3, 4, 4], h)
    scan = rows.tobytes()
    plte = bytes(rng.integers(0, 256, 3 * 256, dtype=np.uint8)) if ctype == 3 else None
    got = np.zeros((h, w, 4), np.uint8)
    src = (L.PngSource * 1)(L.PngSource(scan, w, h, ctype, depth, plte, 256 if plte else 0, None, 0, got.ctypes.data))
    best = 1e9
    for rep in range(3):
        t = time.perf_counter(); rc = lib.pngloss_hip_png_decode_batch_host(ctx._ctx, src, 1); best = min(best, time.perf_counter() - t)
        assert rc == 0
    print("colour type %d depth %2d (%d bytes per pixel group): %7.1f ms per call, %6.1f Mpx/s" % (ctype, depth, max(1, channels * depth // 8), best * 1e3, w * h / best / 1e6))
