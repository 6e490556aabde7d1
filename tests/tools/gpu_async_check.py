"""Check + timing aid (not a test): pngloss_hip_optimize_batch_async on one 4096x4096 frame -- how long the call takes to return, how much
host CPU time the process spends while the batch runs (getrusage: all threads, the launch thread included), whether host work between
_async and _finish overlaps the GPU, digests.   usage: gpu_async_check.py [W H [s b]]"""
import os
import resource
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pngloss_amd as P  # noqa: E402
import torch  # noqa: E402

w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
s = int(sys.argv[3]) if len(sys.argv) > 3 else 19
b = int(sys.argv[4]) if len(sys.argv) > 4 else 2
img = P.synth_rgba(w, h, 0, 0)
ctx = P.HipContext()


def cpu_s():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


for rep in range(3):
    d = torch.from_numpy(img.copy()).cuda()
    f = torch.zeros(h, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    c0, t0 = cpu_s(), time.perf_counter()
    ctx.enqueue([(d.data_ptr(), f.data_ptr(), w, h)], s, b, stream=torch.cuda.current_stream().cuda_stream)
    t1 = time.perf_counter()
    # host work that wants to overlap: a sleep (a busy host would show as CPU time)
    time.sleep(0.05)
    t2 = time.perf_counter()
    res = ctx.finish()
    t3, c3 = time.perf_counter(), cpu_s()
    out = d.cpu().numpy()
    print(f"{w}x{h} s={s} b={b}: _async returned after {1e3 * (t1 - t0):.2f} ms; _finish {1e3 * (t3 - t2):.1f} ms after 50 ms of host work; wall {1e3 * (t3 - t0):.1f} ms, "
          f"engine {ctx.engine_ms:.1f} ms; host CPU {1e3 * (c3 - c0):.1f} ms  {ctx.engine_info(0)}  out={P.fnv1a64(out, P.SURVEY_FNV_BASIS):016x}", flush=True)
