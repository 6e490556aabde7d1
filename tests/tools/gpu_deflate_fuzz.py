#!/usr/bin/env python3
"""Fuzz the GPU deflate against the CPU run of the same encoder: random shapes, contents, strengths and batch mixes;
every zlib stream must (a) inflate to the emitted scanlines and (b) equal the CPU stream byte for byte.
  python tests/tools/gpu_deflate_fuzz.py SECONDS [SEED]"""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import pngloss_amd as P  # noqa: E402
from tests import util as U  # noqa: E402


def content(rng, h, w):
    kind = rng.integers(0, 8)
    if kind == 0:
        a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    elif kind == 1:
        a = P.synth_rgba(w, h, int(rng.integers(0, 6)), int(rng.integers(0, 1000)))
    elif kind == 2:
        a = np.full((h, w, 4), rng.integers(0, 256), np.uint8)
    elif kind == 3:                                   # few-valued, long repeats
        a = rng.integers(0, 3, (h, w, 4), dtype=np.uint8) * 80
    elif kind == 4:                                   # horizontal stripes -> matches at distance = stride
        a = np.repeat(rng.integers(0, 256, (h, 1, 4), dtype=np.uint8), w, axis=1)
    elif kind == 5:                                   # periodic texture
        t = rng.integers(0, 256, (1 + h // 7, 1 + w // 5, 4), dtype=np.uint8)
        a = np.tile(t, (8, 6, 1))[:h, :w]
    elif kind == 6:                                   # gray, opaque
        g = rng.integers(0, 256, (h, w), dtype=np.uint8)
        a = np.stack([g, g, g, np.full_like(g, 255)], axis=2)
    else:                                             # smooth gradient + transparent holes
        y, x = np.mgrid[0:h, 0:w]
        a = np.stack([(x * 3) & 255, (y * 2) & 255, (x + y) & 255, np.where((x // 8 + y // 8) % 3 == 0, 0, 255)], axis=2).astype(np.uint8)
    return np.ascontiguousarray(a)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctx = P.HipContext(0)
    t0 = time.time()
    images = bad = 0
    while time.time() - t0 < seconds:
        n = int(rng.integers(1, 7))
        arrays = []
        for _ in range(n):
            big = rng.random() < 0.15
            h = int(rng.integers(1, 700 if big else 60))
            w = int(rng.integers(1, 900 if big else 120))
            arrays.append(content(rng, h, w))
        s = int(rng.choice([0, 5, 19, 40, 85, 255]))
        b = int(rng.choice([1, 2, 8, 32767]))
        wf = bool(rng.integers(0, 2))
        outs, filts, emitted = ctx.run_host_emit(arrays, s, b, wf)
        outs2, filts2, streams = ctx.run_host_zlib(arrays, s, b, wf)
        for i in range(n):
            ctype, ids, rows = emitted[i]
            want = np.concatenate([ids[:, None], rows], axis=1).tobytes()
            zc, z, blocks = streams[i]
            ok = zc == ctype and zlib.decompress(z) == want and z == U.deflate_host(want, team=0)[0]
            images += 1
            if not ok:
                bad += 1
                print(f"MISMATCH seed {seed}: image {i} of batch, shape {arrays[i].shape}, s={s} b={b} filters={wf}", flush=True)
    print(f"deflate fuzz: {images} images in {time.time() - t0:.0f} s, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
