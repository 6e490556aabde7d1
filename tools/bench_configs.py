"""Throughput + bit-exactness of the remaining BASELINE.json configurations on ONE MI355X (configs[2], [3], [4]).
Not the headline (bench.py is); results are recorded under profiles/.   python tools/bench_configs.py [--skip-sweep]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pngloss_amd as P  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIG = json.load(open(os.path.join(ROOT, "tests", "golden", "digests.json")))["synthetic"]


def digest(a):
    return "%016x" % P.fnv1a64(a, P.SURVEY_FNV_BASIS)


def run_batch(ctx, imgs, s, b, stream=0):
    dev = [torch.from_numpy(a).cuda() for a in imgs]
    filt = [torch.zeros(a.shape[0], dtype=torch.uint8, device="cuda") for a in imgs]
    torch.cuda.synchronize()
    t = time.perf_counter()
    res = ctx.run([(d.data_ptr(), f.data_ptr(), a.shape[1], a.shape[0]) for d, f, a in zip(dev, filt, imgs)], s, b, stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    return dev, filt, res, dt, ctx.engine_ms


def main():
    out = {}
    ctx = P.HipContext()
    # ---- configs[2]: the 11 suite images (sizes and byte-per-pixel classes of SURVEY section 4; synthetic stand-ins,
    #      the PNG files cannot travel to the GPU box), one image-parallel batch
    suite = [("barbara", 512, 512, 4), ("david", 180, 215, 4), ("ssr", 900, 645, 4), ("girl", 755, 503, 2),
             ("lena", 512, 512, 2), ("parrots", 768, 512, 2), ("rose", 70, 46, 2), ("tenko", 554, 382, 2),
             ("dice", 800, 600, 5), ("redbrush", 512, 480, 5), ("tux", 265, 314, 5)]
    imgs = [P.synth_rgba(w, h, m, i) for i, (_, w, h, m) in enumerate(suite)]
    run_batch(ctx, imgs, 19, 2)
    dev, filt, res, dt, eng = run_batch(ctx, imgs, 19, 2)
    px = sum(a.shape[0] * a.shape[1] for a in imgs)
    out["configs[2] suite-shaped batch (11 images, 2.94 Mpx, classes 1/3/4)"] = dict(
        mpx_per_s=round(px / dt / 1e6, 2), seconds=round(dt, 4), engine_ms=round(eng, 2), bpp=[r["bpp"] for r in res])
    # ---- configs[3]: 256 frames 1920x1080 mode 0 (one GPU's view of the 8-GPU job is 32 frames; both are measured)
    for nframes in (32, 256):
        frames = [P.synth_rgba(1920, 1080, 0, f) for f in range(nframes)]
        dev, filt, res, dt, eng = run_batch(ctx, frames, 19, 2)
        ok = True
        for f in (0, 1, 255):
            if f < nframes:
                e = [e for e in DIG if e["width"] == 1920 and e["frame"] == f][0]
                ok &= digest(dev[f].cpu().numpy()) == e["out"] and digest(filt[f].cpu().numpy()) == e["filters"]
        out[f"configs[3] {nframes} x 1920x1080 frames, one batch on one GPU"] = dict(
            mpx_per_s=round(nframes * 1920 * 1080 / dt / 1e6, 2), seconds=round(dt, 4), engine_ms=round(eng, 2),
            digests_match_reference=bool(ok), repaired_pixels_frame0=res[0]["repaired_pixels"])
        del dev, filt, frames
    # ---- configs[4]: 8192x8192 sweep, the 12 (s, b) points run concurrently (one workgroup = one CU each)
    if "--skip-sweep" not in sys.argv:
        base = P.synth_rgba(8192, 8192, 0, 0)
        points = [(s, b) for s in (0, 20, 40, 85) for b in (1, 2, 8)]
        ctxs = [P.HipContext() for _ in points]
        streams = [torch.cuda.Stream() for _ in points]
        dev = [torch.from_numpy(base).cuda() for _ in points]
        filt = [torch.zeros(8192, dtype=torch.uint8, device="cuda") for _ in points]
        torch.cuda.synchronize()
        t = time.perf_counter()
        for c, st, d, f, (s, b) in zip(ctxs, streams, dev, filt, points):
            c.enqueue([(d.data_ptr(), f.data_ptr(), 8192, 8192)], s, b, stream=st.cuda_stream)
        for c in ctxs:
            c.finish()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t
        sweep = {}
        for c, d, f, (s, b) in zip(ctxs, dev, filt, points):
            e = [e for e in DIG if e["width"] == 8192 and e["strength"] == s and (e["bleed"] == b or s == 0)][0]
            sweep[f"s={s} b={b}"] = dict(engine_s=round(c.engine_ms / 1e3, 2), mpx_per_s=round(67.108864 / (c.engine_ms / 1e3), 3),
                                         out_ok=digest(d.cpu().numpy()) == e["out"], filters_ok=digest(f.cpu().numpy()) == e["filters"])
        out["configs[4] 8192x8192 sweep (12 points concurrently on 12 CUs)"] = dict(wall_s=round(wall, 2), points=sweep)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
