#!/usr/bin/env python3
"""Dump the scanline streams (filter byte + filtered row, what the PNG encoder deflates) of optimised images to
raw files, for experiments with tests/c/deflate_host.  Uses the CPU oracle, so this is a development tool only.
  python tests/tools/dump_streams.py OUTDIR [png ...]      (without PNGs: synthetic frames of all six modes)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import util  # noqa: E402


def stream_of(rgba, strength=19, bleed=2):
    out, flags = util.run_port(rgba, strength, bleed, True)
    ctype, ids, rows = util.png_scanlines_reference(out, flags)
    return np.concatenate([ids[:, None], rows], axis=1).tobytes(), ctype


def main():
    outdir = sys.argv[1]
    os.makedirs(outdir, exist_ok=True)
    pngs = sys.argv[2:]
    if pngs:
        from PIL import Image
        for p in pngs:
            rgba = np.array(Image.open(p).convert("RGBA"))
            data, ctype = stream_of(rgba)
            name = os.path.join(outdir, os.path.splitext(os.path.basename(p))[0] + ".raw")
            open(name, "wb").write(data)
            print(name, rgba.shape, "ctype", ctype, len(data))
    else:
        import pngloss_amd as P
        for mode in range(6):
            rgba = P.synth_rgba(512, 384, mode, 0)
            data, ctype = stream_of(rgba)
            name = os.path.join(outdir, f"synth_m{mode}.raw")
            open(name, "wb").write(data)
            print(name, "ctype", ctype, len(data))


if __name__ == "__main__":
    main()
