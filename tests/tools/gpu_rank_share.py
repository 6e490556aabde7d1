"""Timing aid (not a test): the first n frames of BASELINE.json configs[3] (1920x1080, generator mode 0, s=19 b=2) as ONE device-resident batch through the synchronous entry
point, launch groups opted in like bench.py's batch legs; every frame checked against its reference digest.  usage: gpu_rank_share.py n [n ...]   (env: the library's hooks,
e.g. PNGLOSS_HIP_SEG_SEEDS=0, PNGLOSS_HIP_ENGINE=seg; SHARE_REPS=2)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

import pngloss_amd as P  # noqa: E402

W, H = 1920, 1080
ns = [int(v) for v in sys.argv[1:]] or [32, 64]
want = {e["frame"]: e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "digests_1080p.json")))["frames"]}
base = [torch.from_numpy(P.synth_rgba(W, H, 0, i)).cuda() for i in range(max(ns))]
ctx = P.HipContext()
ctx.set_option("launch_groups", os.environ.get("SHARE_GROUPS", "3"))
for n in ns:
    best = None
    for rep in range(int(os.environ.get("SHARE_REPS", "2"))):
        dev = [b.clone() for b in base[:n]]
        flt = [torch.zeros(H, dtype=torch.uint8, device="cuda") for _ in dev]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ctx.run([(d.data_ptr(), f.data_ptr(), W, H) for d, f in zip(dev, flt)], 19, 2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if best is None or dt < best[0]:
            infos = [ctx.engine_info(i) for i in range(n)]
            best = (dt, ctx.engine_ms, dev, flt, res, infos)
    dt, eng, dev, flt, res, infos = best
    ok = all(r["status"] == 0 for r in res) and all("%016x" % P.fnv1a64(dev[i].cpu().numpy(), P.SURVEY_FNV_BASIS) == want[i]["out"] and
                                                   "%016x" % P.fnv1a64(flt[i].cpu().numpy(), P.SURVEY_FNV_BASIS) == want[i]["filters"] for i in range(n))
    print("n=%3d  wall %8.2f ms  engine %8.2f ms  %7.1f Mpx/s  %s  groups %s  attempts max %d  breaks/walked segments: sum %d max %d  digests %s" % (
        n, dt, eng, n * W * H / dt / 1e3, infos[0]["engine"], infos[0].get("launch_groups"), max(i["attempts"] for i in infos),
        sum(i.get("walked_segments", 0) for i in infos), max(i.get("walked_segments", 0) for i in infos), "ok" if ok else "MISMATCH"), flush=True)
    del dev, flt
ctx.close()
