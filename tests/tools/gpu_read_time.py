"""Read side on the device, timed: n synthetic W x H RGBA files (generator mode 0, PNG filter "sub" on every row, zlib level 6) through
  (a) host zlib inflate on `threads` threads + pngloss_hip_png_decode_batch_device   (what the command line tool's --gpu-read does, minus its download)
  (b) pngloss_hip_png_decode_batch_device_z   (inflate on the device too)
usage: gpu_read_time.py [n] [W] [H] [threads]"""
import os, struct, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pngloss_amd as P
from pngloss_amd import lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
H = int(sys.argv[3]) if len(sys.argv) > 3 else 720
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 16


def chunk(tag, body):
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)


def png_of(img):
    h, w, _ = img.shape
    rows = img.reshape(h, w * 4).astype(np.int16)
    sub = rows.copy(); sub[:, 4:] -= rows[:, :-4]
    raw = np.concatenate([np.full((h, 1), 1, np.uint8), (sub & 255).astype(np.uint8)], axis=1).tobytes()
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


# READ_DISTINCT=k: only k distinct files, repeated (the aggregate of many streams without compressing hundreds of frames first)
distinct = min(n, int(os.environ.get("READ_DISTINCT", n)))
frames = [P.synth_rgba(W, H, 0, i) for i in range(distinct)]
files = [png_of(f) for f in frames]
files = [files[i % distinct] for i in range(n)]
zbytes = sum(len(L.parse_png(f)["zstream"]) for f in files[:1]) * n
ctx = P.HipContext()
ctx.png_decode_device(files[:2])                      # runtime, code objects
for rep in range(3):
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        parsed = list(ex.map(L.parse_png, files))     # chunk walk + zlib inflate (releases the GIL)
    t1 = time.perf_counter()
    fr, st = ctx.png_decode_device(files, pinned=False)   # (parses again inside: subtract)
    t2 = time.perf_counter()
    fr2, st2, rc = ctx.png_decode_device_z(files)
    t3 = time.perf_counter()
    print("%d x %dx%d (%.1f MB of scanlines, %.1f MB compressed): host inflate on %d threads %.1f ms; device decode from scanlines (incl. a second host inflate by the wrapper) %.1f ms; "
          "device inflate + decode (incl. the wrapper's host inflate) %.1f ms" % (n, W, H, n * H * (W * 4 + 1) / 1e6, zbytes / 1e6, threads, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
# the library calls alone: prepared sources
import ctypes as C
parsed = [L.parse_png(f) for f in files]
lib = P.hip_lib()
src = (L.PngSource * n)(); zsrc = (L.PngZSource * n)()
for i, p in enumerate(parsed):
    src[i] = L.PngSource(p["scanlines"], W, H, 6, 8, None, 0, None, 0, None)
    zsrc[i] = L.PngZSource(p["zstream"], len(p["zstream"]), W, H, 6, 8, None, 0, None, 0)
ptrs = (C.c_void_p * n)(); st = (C.c_int * n)()
lib.pngloss_hip_png_decode_batch_device.argtypes = [C.c_void_p, C.POINTER(L.PngSource), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]
lib.pngloss_hip_png_decode_batch_device_z.argtypes = [C.c_void_p, C.POINTER(L.PngZSource), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]
for rep in range(3):
    t0 = time.perf_counter(); rc1 = lib.pngloss_hip_png_decode_batch_device(ctx._ctx, src, n, ptrs, st, None)
    t1 = time.perf_counter(); rc2 = lib.pngloss_hip_png_decode_batch_device_z(ctx._ctx, zsrc, n, ptrs, st, None)
    t2 = time.perf_counter()
    print("library calls alone: _device (pageable scanlines up + inverse filters + expansion) %.1f ms rc %d; _device_z (compressed bytes up + inflate + the rest) %.1f ms rc %d = %.1f MB/s of scanlines per stream"
          % ((t1 - t0) * 1e3, rc1, (t2 - t1) * 1e3, rc2, H * (W * 4 + 1) / 1e6 / max(1e-9, (t2 - t1) - (t1 - t0) * 0.3)), flush=True)
ctx.close()
