"""Timing aid (not a test): a batch of real photographs (the reference's suite: lena, barbara, girl, parrots, tenko -- tiled to n images) on the segment engine, units / segments from
seeds against from every state; bytes compared between the runs.  usage: gpu_photo_batch.py n [n ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pngloss_amd as P

z = np.load(os.path.join(ROOT, "tests", "golden", "suite_inputs.npz"))
photos = [np.ascontiguousarray(z[k]) for k in ("lena", "barbara", "girl", "parrots", "tenko")]
for n in [int(v) for v in sys.argv[1:]] or [24, 64]:
    imgs = [photos[i % len(photos)] for i in range(n)]
    ref = None
    for seeds in ("0", "1"):
        os.environ["PNGLOSS_HIP_SEG_SEEDS"] = seeds
        os.environ["PNGLOSS_HIP_ENGINE"] = "seg"
        ctx = P.HipContext(); ctx.set_option("launch_groups", "3")
        best = None
        for rep in range(2):
            dev = [torch.from_numpy(a.copy()).cuda() for a in imgs]
            flt = [torch.zeros(a.shape[0], dtype=torch.uint8, device="cuda") for a in imgs]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = ctx.run([(d.data_ptr(), f.data_ptr(), a.shape[1], a.shape[0]) for d, f, a in zip(dev, flt, imgs)], 19, 2)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
            if best is None or dt < best: best = dt
        infos = [ctx.engine_info(i) for i in range(n)]
        got = [d.cpu().numpy() for d in dev[:5]]
        same = True if ref is None else all(np.array_equal(a, b) for a, b in zip(ref, got))
        ref = ref or got
        segs = sum((a.shape[1] + 31) // 32 for a in imgs)
        print("n=%3d photographs (%d segments)  PNGLOSS_HIP_SEG_SEEDS=%s  wall %8.2f ms  %6.1f Mpx/s  attempts max %d  breaks sum %d max %d  same bytes as the other run: %s" % (
            n, segs, seeds, best, sum(a.shape[0] * a.shape[1] for a in imgs) / best / 1e3, max(i["attempts"] for i in infos), sum(i["walked_segments"] for i in infos), max(i["walked_segments"] for i in infos), same), flush=True)
        ctx.close()
