"""Timing aid (not a test): BASELINE.json configs[3] through pngloss_hip_multi with EIGHT contexts on device 0 ("0,0,0,0,0,0,0,0": eight host threads, eight launch threads, one GPU) --
what a one-GPU box can show of a node's host side: wall time, the process's CPU time, the CPUs its threads ran on.  usage: gpu_multi8.py [frames] [reps]   (env: PNGLOSS_HIP_PIN=0)"""
import os, resource, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402,F401
import pngloss_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
imgs = [P.synth_rgba(1920, 1080, 0, i) for i in range(n)]
multi = P.HipMulti("0,0,0,0,0,0,0,0")
best = None
for rep in range(reps):
    r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
    outs, filts, res = multi.run_host(imgs, 19, 2)
    dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
    assert all(r["status"] == 0 for r in res)
    if best is None or dt < best[0]: best = (dt, cpu)
print("eight contexts on one device, %d frames of 1920x1080 from host memory (PNGLOSS_HIP_PIN=%s; affinity set %d CPUs): wall %.1f ms (%.0f Mpx/s incl. staging and PCIe), process CPU time %.1f ms"
      % (n, os.environ.get("PNGLOSS_HIP_PIN", "default"), len(os.sched_getaffinity(0)), best[0] * 1e3, n * 1920 * 1080 / best[0] / 1e6, best[1] * 1e3))
multi.close()
