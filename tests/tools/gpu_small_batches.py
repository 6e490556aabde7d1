"""Timing aid (not a test): small and mixed batches on the library's own engine choice, in ONE process (like bench.py's legs): the reference's suite as one batch,
a few 1080p frames, frames of mixed heights -- with the launch groups the library picks and with PNGLOSS_HIP_SEG_GROUPS=1 (one sequence for all), interleaved;
and a single frame at the end of every round (a process-wide slowdown of the engine would show there).  usage: gpu_small_batches.py [rounds]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pngloss_amd as P

z = np.load(os.path.join(ROOT, "tests", "golden", "suite_inputs.npz"))
suite = [np.ascontiguousarray(z[k]) for k in sorted(z.files)]
f1080 = [P.synth_rgba(1920, 1080, 0, i) for i in range(4)]
mixed = [P.synth_rgba(1920, 1080, 0, 0), P.synth_rgba(1280, 720, 0, 1), P.synth_rgba(1280, 720, 0, 2), P.synth_rgba(800, 600, 0, 3), P.synth_rgba(2048, 256, 0, 4)]
single = [P.synth_rgba(4096, 1024, 0, 0)]
cases = [("suite (11 images)", suite), ("2 x 1080p", f1080[:2]), ("4 x 1080p", f1080), ("6 x 1080p", f1080 + f1080[:2]), ("mixed (1080p, 2 x 720p, 800x600, 2048x256)", mixed), ("one 4096x1024", single)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
res = {}
digests = {}
for rnd in range(rounds):
    for groups in ("default", "1"):
        if groups == "default": os.environ.pop("PNGLOSS_HIP_SEG_GROUPS", None)
        else: os.environ["PNGLOSS_HIP_SEG_GROUPS"] = groups
        for name, imgs in cases:
            ctx = P.HipContext()
            dev = [torch.from_numpy(a.copy()).cuda() for a in imgs]
            filt = [torch.zeros(a.shape[0], dtype=torch.uint8, device="cuda") for a in imgs]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = ctx.run([(d.data_ptr(), f.data_ptr(), a.shape[1], a.shape[0]) for d, f, a in zip(dev, filt, imgs)], 19, 2)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            dg = tuple(P.fnv1a64(d.cpu().numpy(), P.SURVEY_FNV_BASIS) for d in dev) + tuple(P.fnv1a64(f.cpu().numpy(), P.SURVEY_FNV_BASIS) for f in filt)
            if name in digests and digests[name] != dg: print("DIGESTS DIFFER:", name, groups)
            digests.setdefault(name, dg)
            key = (name, groups)
            res[key] = min(res.get(key, 1e9), dt)
            ctx.close()
for name, imgs in cases:
    px = sum(a.shape[0] * a.shape[1] for a in imgs)
    a, b = res[(name, "default")], res[(name, "1")]
    print("%-46s library's groups %8.2f ms (%6.1f Mpx/s)   one group %8.2f ms (%6.1f Mpx/s)" % (name, a, px / a / 1e3, b, px / b / 1e3))
