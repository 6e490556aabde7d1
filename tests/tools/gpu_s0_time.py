import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import pngloss_amd as P
for (w, h, n) in [(8192, 8192, 1), (4096, 4096, 1), (1920, 1080, 1), (1920, 1080, 64)]:
    imgs = [P.synth_rgba(w, h, 0, i % 4) for i in range(min(n, 4))]
    for eng in (None, "seg"):
        if eng: os.environ["PNGLOSS_HIP_ENGINE"] = eng
        else: os.environ.pop("PNGLOSS_HIP_ENGINE", None)
        if eng == "seg" and n > 1: continue
        ctx = P.HipContext()
        best = None
        for rep in range(3):
            dev = [torch.from_numpy(imgs[i % len(imgs)].copy()).cuda() for i in range(n)]
            filt = [torch.zeros(h, dtype=torch.uint8, device="cuda") for _ in range(n)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.run([(d.data_ptr(), f.data_ptr(), w, h) for d, f in zip(dev, filt)], 0, 2)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        print("%dx%d n=%d engine %-28s wall %8.2f ms  engine %8.2f ms  %8.1f Mpx/s  filters=%016x" % (w, h, n, ctx.engine_info(0)["engine"], best, ctx.engine_ms, n * w * h / best / 1e3, P.fnv1a64(filt[0].cpu().numpy(), P.SURVEY_FNV_BASIS)))
        ctx.close()
