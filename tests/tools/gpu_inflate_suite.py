"""The device inflate on the reference's eleven suite files (tests/golden/suite_png.npz), ONE stream per call: MB/s of scanlines per stream by the kind of content,
next to what the stream consists of (literals, matches: counted by the decoder's body on the CPU, tests/c/inflate_host.cpp) and zlib on one host thread.
usage: gpu_inflate_suite.py"""
import ctypes as C, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pngloss_amd as P
from pngloss_amd import lib as L
from tests import util as U
import tests.test_inflate_host as T

host = T.inflate_lib()
host.inflate_host_stats.argtypes = [C.c_void_p]
s = U.load_npz("suite_png.npz")
ctx = P.HipContext()
lib = P.hip_lib()
lib.pngloss_hip_png_decode_batch_device_z.argtypes = [C.c_void_p, C.POINTER(L.PngZSource), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]
lib.pngloss_hip_png_decode_batch_device.argtypes = [C.c_void_p, C.POINTER(L.PngSource), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]
print("file                  pixels   scanline bytes  compressed   literals    matches (bytes)      device inflate        zlib, one host thread")
tot_b = 0; tot_t = 0.0
for k in sorted(s.files):
    png = s[k].tobytes()
    p = L.parse_png(png)
    z, raw = p["zstream"], p["scanlines"]
    st8 = (C.c_ulonglong * 8)(); host.inflate_host_stats(st8)
    rc, out = T.run(z, len(raw)); assert rc == 0 and out == raw
    host.inflate_host_stats(st8)
    zsrc = (L.PngZSource * 1)(); src = (L.PngSource * 1)()
    zsrc[0] = L.PngZSource(z, len(z), p["width"], p["height"], p["ctype"], p["depth"], p["plte"], len(p["plte"]) // 3 if p["plte"] else 0, p["trns"], len(p["trns"]) if p["trns"] else 0)
    src[0] = L.PngSource(raw, p["width"], p["height"], p["ctype"], p["depth"], p["plte"], len(p["plte"]) // 3 if p["plte"] else 0, p["trns"], len(p["trns"]) if p["trns"] else 0, None)
    ptrs = (C.c_void_p * 1)(); st = (C.c_int * 1)()
    best_z = best_d = 1e9
    for rep in range(4):
        t0 = time.perf_counter(); rc1 = lib.pngloss_hip_png_decode_batch_device(ctx._ctx, src, 1, ptrs, st, None)
        t1 = time.perf_counter(); rc2 = lib.pngloss_hip_png_decode_batch_device_z(ctx._ctx, zsrc, 1, ptrs, st, None)
        t2 = time.perf_counter()
        assert rc1 == 0 and rc2 == 0 and st[0] == 0
        best_d = min(best_d, t1 - t0); best_z = min(best_z, t2 - t1)
    t0 = time.perf_counter()
    for rep in range(5): zlib.decompress(z)
    tz = (time.perf_counter() - t0) / 5
    infl = max(1e-9, best_z - best_d)          # (the same call without the inflate: upload of the scanlines + inverse filters + expansion)
    tot_b += len(raw); tot_t += infl
    print("%-20s %8d %12d %12d %10d %10d (%9d)   %7.2f ms = %5.1f MB/s   %6.2f ms = %5.0f MB/s"
          % (k, p["width"] * p["height"], len(raw), len(z), st8[2], st8[3], st8[4], infl * 1e3, len(raw) / 1e6 / infl, tz * 1e3, len(raw) / 1e6 / tz), flush=True)
print("all eleven, one after the other: %.1f MB/s of scanlines per stream" % (tot_b / 1e6 / tot_t))
ctx.close()
