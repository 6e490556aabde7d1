"""Diagnostic (not a test): the shader clock the driver reports (hwmon freq1_input / pp_dpm_sclk) sampled from a host thread while the row engines run: one image on the
workgroup-per-image engine, then 8, then 64; one 4096x1024 frame on the segment engine.  usage: gpu_clock_probe.py"""
import glob, os, sys, threading, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pngloss_amd as P

def find():
    c = []
    for d in glob.glob("/sys/class/drm/card*/device"):
        for f in glob.glob(d + "/hwmon/hwmon*/freq1_input"): c.append(f)
    return c
files = find()
print("clock files:", files)
for d in glob.glob("/sys/class/drm/card*/device"):
    for n in ("pp_dpm_sclk", "power_dpm_force_performance_level"):
        try: print(d, n, open(os.path.join(d, n)).read().strip().replace("\n", " | "))
        except OSError as e: print(d, n, "unreadable", e)
stop = False
samples = []
def sampler():
    while not stop:
        v = []
        for f in files:
            try: v.append(int(open(f).read()) // 1000000)
            except (OSError, ValueError): pass
        samples.append((time.perf_counter(), v))
        time.sleep(0.002)
ctx = P.HipContext()
def run(eng, w, h, n):
    global stop, samples
    os.environ["PNGLOSS_HIP_ENGINE"] = eng
    imgs = [P.synth_rgba(w, h, 0, i % 4) for i in range(n)]
    dev = [torch.from_numpy(a.copy()).cuda() for a in imgs]; filt = [torch.zeros(h, dtype=torch.uint8, device="cuda") for _ in imgs]
    torch.cuda.synchronize()
    stop = False; samples = []
    t = threading.Thread(target=sampler); t.start()
    time.sleep(0.02)
    t0 = time.perf_counter()
    ctx.run([(d.data_ptr(), f.data_ptr(), w, h) for d, f in zip(dev, filt)], 19, 2)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    stop = True; t.join()
    inside = [v for (ts, v) in samples if t0 <= ts <= t1 and v]
    flat = sorted(x for v in inside for x in v)
    print("%-3s %dx%d n=%-3d engine %8.2f ms   sclk MHz during the run: min %s median %s max %s (%d samples)" % (eng, w, h, n, ctx.engine_ms, flat[0] if flat else None, flat[len(flat) // 2] if flat else None, flat[-1] if flat else None, len(inside)))
for rep in range(2):
    run("wg", 1920, 1080, 1); run("wg", 1920, 1080, 2); run("wg", 1920, 1080, 4); run("wg", 1920, 1080, 8); run("wg", 1920, 1080, 64); run("wg", 1920, 1080, 256)
    run("seg", 4096, 1024, 1); run("seg", 1920, 1080, 32)
