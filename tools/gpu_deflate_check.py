#!/usr/bin/env python3
"""GPU deflate smoke/measurement: zlib streams from the device must inflate to exactly the emitted scanlines; prints
sizes next to zlib level 9 / Z_FILTERED (what libpng gives the reference) and timings.
  python tools/gpu_deflate_check.py [quick|full]"""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pngloss_amd as P  # noqa: E402


def zlib9f(data):
    c = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FILTERED)
    return c.compress(data) + c.flush()


def check(ctx, arrays, label, time_zlib=True):
    t0 = time.perf_counter()
    outs, filts, emitted = ctx.run_host_emit(arrays)
    t1 = time.perf_counter()
    outs2, filts2, streams = ctx.run_host_zlib(arrays)
    t2 = time.perf_counter()
    dms = ctx.deflate_ms
    tot_in = tot_gpu = tot_z = 0
    tz = 0.0
    ok = True
    for i in range(len(arrays)):
        ctype, ids, rows = emitted[i]
        want = np.concatenate([ids[:, None], rows], axis=1).tobytes() if rows.size else b""
        zc, zbytes, blocks = streams[i]
        got = zlib.decompress(zbytes) if zbytes else b""
        if got != want or zc != ctype or not np.array_equal(outs[i], outs2[i]):
            ok = False
            print(f"  MISMATCH image {i}: ctype {zc} vs {ctype}, inflated {len(got)} vs {len(want)} bytes")
        tot_in += len(want)
        tot_gpu += len(zbytes)
        if time_zlib:
            t = time.perf_counter()
            tot_z += len(zlib9f(want))
            tz += time.perf_counter() - t
    ratio = tot_gpu / tot_z if tot_z else float("nan")
    print(f"{label}: {len(arrays)} images, {tot_in} scanline bytes -> gpu {tot_gpu} B, zlib9f {tot_z} B (gpu/zlib {ratio:.4f}); "
          f"emit call {1e3 * (t1 - t0):.0f} ms, zlib call {1e3 * (t2 - t1):.0f} ms (deflate stage {dms:.1f} ms), "
          f"cpu zlib {1e3 * tz:.0f} ms; {'OK' if ok else 'FAILED'}", flush=True)
    return ok


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    ctx = P.HipContext(0)
    ok = True
    ok &= check(ctx, [P.synth_rgba(64, 48, m, 0) for m in range(6)], "tiny all modes")
    ok &= check(ctx, [P.synth_rgba(w, h, 0, 1) for (w, h) in [(1, 1), (2, 3), (5, 1), (1, 7), (300, 1), (1, 300)]], "edge shapes")
    ok &= check(ctx, [P.synth_rgba(512, 384, m, 0) for m in range(6)], "512x384 all modes")
    ok &= check(ctx, [P.synth_rgba(1920, 1080, 0, 0)], "1080p x1")
    if mode == "full":
        ok &= check(ctx, [P.synth_rgba(1920, 1080, 0, f) for f in range(16)], "1080p x16")
        ok &= check(ctx, [P.synth_rgba(4096, 4096, 0, 0)], "4096^2 x1")
        ok &= check(ctx, [P.synth_rgba(1920, 1080, 1, 0)], "1080p noise")
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
