#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

  metric   : Mpixels/s of the filter+quantise hot path (optimize_with_rows semantics), bit-exact vs the reference
  workload : BASELINE.json configs[1] -- ONE 4096x4096 synthetic RGBA8 frame (SURVEY.md Appendix B generator,
             mode 0 "photo"), --strength 19 --bleed 2, per GPU.  A "step" is one pass of the whole hot path
             (classify -> original histograms -> row engine -> [unpack]) over that frame, input already resident
             in HBM, output pixels + filter IDs left in HBM.
  N > 1    : `value` is WEAK scaling of independent frames -- every rank optimises its own 4096x4096 frame (frame
             index = rank) with no data-path collective; RCCL only carries the barrier and the gather of the per-image
             result records.  value = N * pixels * steps / max-over-ranks(time).
  batch    : for every N, additionally BASELINE.json configs[3] -- 256 synthetic 1920x1080 frames split over the N ranks
             with pngloss_amd.shard.contiguous_partition, one device-resident batch per rank; reported under the `batch`
             key (whole-job Mpixels/s, per-rank engine ms, reference digests of ALL 256 frames checked) and, as `batch_value`, at the
             top level of the line for every N.  This is the image-batch (strong) scaling north_star asks about; it is not part of `value`.  One GPU runs one image per CU, so
             configs[3] cannot get faster below 256 frames per GPU: `batch_saturating` (512 frames per rank for N <= 8, i.e.
             4096 frames in all at N = 8) is the leg whose rate can scale with N.
  engines  : the library picks the row engine per batch (pngloss_hip_last_engine_info): few large images -> segment-parallel
             (the whole GPU on one image), batches -> one workgroup per image.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (the row engine) against HBM with the
ALGORITHMIC traffic of SURVEY.md section 8(d): 8 bytes per RGBA8 pixel (read 4 + write 4; the H filter bytes are
noise).  `cpu_baseline` is the real reference (oracle/_ref, kind "reference") -- or our restatement (kind "port")
where the prebuilt reference .so is absent -- timed single-threaded on this box's host cores on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, STRENGTH, BLEED, MODE = 4096, 4096, 19, 2, 0
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
ALGO_BYTES_PER_PIXEL = 8
CPU_SAMPLE_ROWS = 1024         # cpu_baseline sample: the top 4096x1024 strip of the same frame
BUILD_CONTAINER_REFERENCE_MPX = 0.515   # the reference, one thread, full 4096x4096 frame, build container
BATCH_FRAMES, BATCH_W, BATCH_H = 256, 1920, 1080      # BASELINE.json configs[3]
def _newest_profile(suffix):
    for r in ("r06", "r05", "r04", "r03"):
        if os.path.exists(os.path.join(ROOT, "profiles", f"{r}_{suffix}")):
            return f"{r}_{suffix}"
    return f"r03_{suffix}"


KERNEL_STATS = _newest_profile("kernel_trace_stats.txt")
PMC_TRAFFIC = _newest_profile("pmc_traffic.json")
RANK_SHARES = (32, 64, 128)     # frames of configs[3] one rank holds at N = 8, 4, 2


def profile_stamp(name):
    """source_digest / head a committed profile was taken at (tools/gpu_round5.sh writes them into the file's first lines; round <= 4 files have none)."""
    out = {"file": "profiles/" + name, "source_digest": None, "head": None}
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            txt = fh.read(4096)
        import re
        m = re.search(r"source_digest[=\":\s]+([0-9a-f]{16})", txt)
        if m:
            out["source_digest"] = m.group(1)
        m = re.search(r"head[=\":\s]+([0-9a-f]{7,40}(?:-dirty)?)", txt)
        if m:
            out["head"] = m.group(1)
    except OSError:
        pass
    return out


def _ref_worker(frame_index):
    """One process of the all-cores CPU baseline: the real reference (or the port) on one 1920x1080 frame."""
    import numpy as np
    import pngloss_amd as P
    sig = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_void_p, C.c_bool, C.c_uint8, C.c_long]
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libpngloss_ref.so")
    if os.path.exists(ref_path):
        lib = C.CDLL(ref_path); fn = lib.optimize_with_rows
    else:
        lib = C.CDLL(os.path.join(ROOT, "oracle", "libpngloss_port.so")); fn = lib.port_optimize_with_rows
    fn.argtypes = sig; fn.restype = C.c_int
    buf = P.synth_rgba(BATCH_W, BATCH_H // 4, MODE, frame_index)          # a quarter-height 1080p frame bounds the leg to a few seconds
    h, w = buf.shape[:2]
    filt = np.zeros(h, np.uint8)
    rows = (C.c_void_p * h)(*[buf.ctypes.data + y * w * 4 for y in range(h)])
    t = time.perf_counter()
    rc = fn(rows, w, h, filt.ctypes.data, False, STRENGTH, BLEED)
    return rc, w * h, time.perf_counter() - t


def _full_frame_worker(q):
    """The REAL reference (or the port), one thread, on the FULL 4096x4096 frame the metric is quoted on (~18 s of host time), in a process of its own while
    this one waits and the GPU idles."""
    import numpy as np
    import pngloss_amd as P
    sig = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_void_p, C.c_bool, C.c_uint8, C.c_long]
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libpngloss_ref.so")
    if os.path.exists(ref_path):
        lib = C.CDLL(ref_path); fn = lib.optimize_with_rows; kind = "reference"
    else:
        lib = C.CDLL(os.path.join(ROOT, "oracle", "libpngloss_port.so")); fn = lib.port_optimize_with_rows; kind = "port"
    fn.argtypes = sig; fn.restype = C.c_int
    buf = P.synth_rgba(W, H, MODE, 0)
    filt = np.zeros(H, np.uint8)
    rows = (C.c_void_p * H)(*[buf.ctypes.data + y * W * 4 for y in range(H)])
    t = time.perf_counter()
    rc = fn(rows, W, H, filt.ctypes.data, False, STRENGTH, BLEED)
    dt = time.perf_counter() - t
    q.put(dict(rc=rc, seconds=dt, kind=kind, out="%016x" % P.fnv1a64(buf, P.SURVEY_FNV_BASIS), filters="%016x" % P.fnv1a64(filt, P.SURVEY_FNV_BASIS)))


def _warm_worker(i):
    import pngloss_amd as P
    P.synth_rgba(64, 8, MODE, i)
    return os.getpid()


def usable_cpus():
    """CPUs this process may really use: the scheduler affinity mask, capped by the cgroup CPU quota (cpu.max) when one is set.
    os.cpu_count() is the host's, not the container's: one process per host core inside a small quota measures the throttle."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota:
        n = max(1, min(n, int(quota)))
    return n, quota


def cpu_all_cores():
    """N processes, one frame strip each (the reference is single-threaded; its natural scale-out is one process per file,
    SURVEY.md 8(d)(ii)), at N = 1, 8, 64 and every usable CPU; per_process_mpx next to each aggregate shows whether the processes
    really ran in parallel."""
    import multiprocessing as mp
    nmax, quota = usable_cpus()
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip(); break
    except OSError:
        pass
    ctxm = mp.get_context("fork")
    curve = []
    for n in sorted({1, min(8, nmax), min(64, nmax), nmax}):
        with ctxm.Pool(n) as pool:
            pool.map(_warm_worker, range(n), chunksize=1)        # process start-up, imports and library loads stay outside the timed map
            t = time.perf_counter()
            res = pool.map(_ref_worker, range(n), chunksize=1)
            wall = time.perf_counter() - t
        assert all(r[0] == 0 for r in res)
        px = sum(r[1] for r in res)
        curve.append({"processes": n, "value": round(px / wall / 1e6, 3), "per_process_mpx": round(sum(r[1] / r[2] for r in res) / len(res) / 1e6, 4), "wall_s": round(wall, 2)})
    best = max(curve, key=lambda c: c["value"])
    return {"value": best["value"], "unit": "Mpixels/s", "cores": best["processes"], "cpu_model": model,
            "usable_cpus": nmax, "cgroup_cpu_quota": quota, "host_logical_cpus": os.cpu_count(),
            "kind": "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libpngloss_ref.so")) else "port",
            "sample": f"one {BATCH_W}x{BATCH_H // 4} strip of a configs[3] frame per process, s={STRENGTH} b={BLEED}",
            "per_process_mpx": best["per_process_mpx"], "curve": curve}


def cpu_baseline(frame0):
    """Time the CPU path on a bounded sample of the same workload (rank 0, N=1 only)."""
    import numpy as np

    sample = np.ascontiguousarray(frame0[:CPU_SAMPLE_ROWS])
    h, w = sample.shape[:2]
    sig = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_void_p, C.c_bool, C.c_uint8, C.c_long]

    def timed(fn):
        buf = sample.copy()
        filt = np.zeros(h, np.uint8)
        rows = (C.c_void_p * h)(*[buf.ctypes.data + y * w * 4 for y in range(h)])
        t = time.perf_counter()
        rc = fn(rows, w, h, filt.ctypes.data, False, STRENGTH, BLEED)
        dt = time.perf_counter() - t
        assert rc == 0
        return w * h / dt / 1e6, buf, filt

    out = {}
    port = C.CDLL(os.path.join(ROOT, "oracle", "libpngloss_port.so"))
    port.port_optimize_with_rows.argtypes = sig
    port.port_optimize_with_rows.restype = C.c_int
    port_mpx, pbuf, pfilt = timed(port.port_optimize_with_rows)
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libpngloss_ref.so")
    kind, value = "port", port_mpx
    if os.path.exists(ref_path):
        ref = C.CDLL(ref_path)
        ref.optimize_with_rows.argtypes = sig
        ref.optimize_with_rows.restype = C.c_int
        value, rbuf, rfilt = timed(ref.optimize_with_rows)
        kind = "reference"
        assert np.array_equal(rbuf, pbuf) and np.array_equal(rfilt, pfilt)
    out = {"value": round(value, 4), "unit": "Mpixels/s", "cores": 1, "kind": kind,
           "sample": f"top {w}x{h} strip of the 4096x4096 frame, s={STRENGTH} b={BLEED}, single thread "
                     f"(the reference is single-threaded); host has {os.cpu_count()} logical cores, {usable_cpus()[0]} usable here",
           "port_value": round(port_mpx, 4),
           "note": f"the build container measured {BUILD_CONTAINER_REFERENCE_MPX} Mpixels/s for the reference on the FULL frame "
                   "(BASELINE.md section 2); the >=50x target of BASELINE.json was defined on that number"}
    try:
        out["all_cores"] = cpu_all_cores()
    except Exception as exc:          # informational only
        out["all_cores"] = {"error": repr(exc)}
    return out


def engine_kernels():
    """Per-kernel share of the row engine from the COMMITTED rocprofv3 kernel trace of this command (static: labelled so in the line)
    (profiles/r0x_kernel_trace_stats.txt): calls, average duration, share of the engine's time."""
    out = {}
    try:
        import csv
        with open(os.path.join(ROOT, "profiles", KERNEL_STATS)) as fh:
            for f in csv.reader(ln for ln in fh if not ln.startswith("#")):
                if len(f) >= 5 and "seg_k_" in f[0] and "resolve" not in f[0]:
                    name = f[0].split("seg_k_")[1].split("(")[0]
                    out["seg_k_" + name] = {"calls": int(f[1]), "avg_us": round(float(f[3]) / 1e3, 2), "percent_of_gpu_time": round(float(f[4]), 2)}
    except (OSError, ValueError, IndexError):
        pass
    if out:
        out["static"] = True
        out["source"] = "profiles/%s (a committed rocprofv3 --kernel-trace --stats run of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-sweep` on the builder's GPU box: NOT measured by this invocation)" % KERNEL_STATS
    return out


def bandwidth_kernels():
    """The HBM-class passes around the row engine, from the committed rocprofv3 kernel trace of this very command
    (profiles/r03_kernel_trace_stats.txt): average duration and bytes moved per launch on the 4096x4096 frame."""
    out = {}
    alg = {"pl_classify": 4 * W * H, "pl_hist": 4 * W * H}   # both read the 4 B/px image once
    try:
        import csv
        with open(os.path.join(ROOT, "profiles", KERNEL_STATS)) as fh:
            for f in csv.reader(ln for ln in fh if not ln.startswith("#")):
                for name, nbytes in alg.items():
                    if len(f) >= 5 and (name + "(" in f[0] or name + "<" in f[0]) and name not in out:
                        avg_us = round(float(f[3]) / 1e3, 2)
                        out[name] = {"avg_us": avg_us, "algorithmic_bytes": nbytes, "GB_per_s": round(nbytes / avg_us / 1e3, 1),
                                     "frac_of_8TBps": round(nbytes / avg_us / 1e3 / 8000.0, 4)}
    except (OSError, ValueError, IndexError):
        pass
    if out:
        out["static"] = True
        out["source"] = "profiles/%s (committed rocprofv3 trace, NOT measured by this invocation; pl_hist is bound by its 20 LDS atomics per pixel, DESIGN.md section 11)" % KERNEL_STATS
    return out


def batch_ctx(P, device):
    """A context for the batch legs: this process only ever uses the SYNCHRONOUS entry point, so it opts into three launch groups for large batches on the
    segment engine (pngloss_hip_set_option "launch_groups" "3": include/pngloss_hip.h; the default of two is for processes that also hand the asynchronous
    entry streams of their own).  Reported as `launch_groups_option` in the line."""
    ctx = P.HipContext(device)
    ctx.set_option("launch_groups", "3")
    global _CALIB_WARMED
    if not _CALIB_WARMED:
        # one-time costs of a process's first batch (the engine's streams and launch thread; with PNGLOSS_HIP_CALIB=1 also the ~30 ms probe that calibrates the engine cost
        # model, pl_host.hip:engine_calib): a warm-up, like the headline's --warmup steps -- two tiny frames here, so that no timed leg carries them
        import torch
        tiny = [torch.from_numpy(P.synth_rgba(96, 8, 0, i)).cuda() for i in range(2)]
        ctx.run([(t.data_ptr(), 0, 96, 8) for t in tiny], STRENGTH, BLEED)
        torch.cuda.synchronize()
        _CALIB_WARMED = True
    return ctx


_CALIB_WARMED = False


def _known_1080p():
    """frame -> reference digests of configs[3] frames (tests/golden/digests_1080p.json: all 256 since round 6; digests.json: 0, 1, 255)"""
    try:
        from tests import util as TU
        return TU.load_digests_1080p()
    except Exception:
        return {}


BATCH_KNOWN = _known_1080p() or {0: None, 1: None, 255: None}


def run_batch(P, S, torch, ctx_factory, rank, world, local_rank, barrier):
    """BASELINE.json configs[3]: 256 x 1920x1080 frames over `world` ranks, one device-resident batch per rank."""
    mine = S.contiguous_partition(BATCH_FRAMES, world)[rank]
    frames = [P.synth_rgba(BATCH_W, BATCH_H, MODE, i) for i in mine]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    filt = [torch.zeros(BATCH_H, dtype=torch.uint8, device="cuda") for _ in frames]
    del frames
    ctx = ctx_factory()
    desc = [(d.data_ptr(), f.data_ptr(), BATCH_W, BATCH_H) for d, f in zip(dev, filt)]
    barrier()
    t0 = time.perf_counter()
    res = ctx.run(desc, STRENGTH, BLEED, stream=torch.cuda.current_stream().cuda_stream) if desc else []
    barrier()
    dt = time.perf_counter() - t0
    eng = ctx.engine_ms if desc else 0.0
    recs = []
    for i, d, f, r in zip(mine, dev, filt, res):
        rec = dict(index=i, status=r["status"])
        if i in BATCH_KNOWN:
            rec["out"] = "%016x" % P.fnv1a64(d.cpu().numpy(), P.SURVEY_FNV_BASIS)
            rec["filters"] = "%016x" % P.fnv1a64(f.cpu().numpy(), P.SURVEY_FNV_BASIS)
        recs.append(rec)
    ctx.close()
    return dt, eng, recs


def run_sweep_8192(P, torch, ctx_factory, golden):
    """BASELINE.json configs[4]: strength {0, 20, 40, 85} x bleed {1, 2, 8} on the 8192x8192 frame, one point after the other on one
    GPU: Mpixels/s (wall clock of enqueue .. finish, input resident), the row engine that ran, its fraction of the HBM roofline
    (8 B/px over the engine's time) and the reference digests of SURVEY.md Appendix B."""
    w = h = 8192
    base = torch.from_numpy(P.synth_rgba(w, h, MODE, 0)).cuda()
    ctx = ctx_factory()
    pts = []
    stream = torch.cuda.current_stream().cuda_stream
    for s in (0, 20, 40, 85):
        for b in (1, 2, 8):
            d = base.clone()
            f = torch.zeros(h, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = ctx.run([(d.data_ptr(), f.data_ptr(), w, h)], s, b, stream=stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            g = [e for e in golden["synthetic"] if e["width"] == w and e["height"] == h and e["strength"] == s and (e["bleed"] == b or s == 0)][0]
            info = ctx.engine_info(0)
            ok = res[0]["status"] == 0 and "%016x" % P.fnv1a64(d.cpu().numpy(), P.SURVEY_FNV_BASIS) == g["out"] and "%016x" % P.fnv1a64(f.cpu().numpy(), P.SURVEY_FNV_BASIS) == g["filters"]
            gbs = ALGO_BYTES_PER_PIXEL * w * h / (ctx.engine_ms * 1e-3) / 1e9
            pts.append({"strength": s, "bleed": b, "value": round(w * h / dt / 1e6, 2), "engine_ms": round(ctx.engine_ms, 1), "engine": info["engine"],
                        "attempts": info["attempts"], "epochs": info["restarts"], "walked_segments": info.get("walked_segments", 0),
                        "roofline_frac": gbs / HBM_PEAK_GBS, "digests_match_reference": bool(ok)})
            del d, f
    ctx.close()
    del base
    torch.cuda.empty_cache()
    return {"workload": "BASELINE.json configs[4]: one 8192x8192 synthetic RGBA8 frame, strength {0,20,40,85} x bleed {1,2,8}, one GPU, point after point",
            "unit": "Mpixels/s", "points": pts, "min_value": min(p["value"] for p in pts), "all_on_segment_engine": all(p["engine"] == "segment-parallel" for p in pts if p["strength"] != 0),
            "strength_0_engine": sorted({p["engine"] for p in pts if p["strength"] == 0}),
            "all_digests_match_reference": all(p["digests_match_reference"] for p in pts),
            "note": "roofline_frac = 8 B/px * 67.1 Mpx / engine time / 8 TB/s: like the headline, bound by the row-to-row dependency, not by HBM; "
                    "strengths whose chain-state set exceeds 1024 (85 at bleed 1 and 2, 40 at bleed 1) run the seeded enumeration (DESIGN.md section 4); "
                    "strength 0 quantises nothing: its points run the row-statistics engine (pl_rows.hip: one parallel pass for every row's residual counts, one serial pass of "
                    "decisions; the image is read twice: roofline_frac there is the fraction of HBM peak that 8 B/px would be)"}


def run_suite_batch(P, torch, ctx_factory, golden):
    """BASELINE.json configs[2]: the reference's eleven suite images (tests/golden/suite_inputs.npz: what its reader makes of suite/*.png)
    as ONE device-resident batch at s=19 b=2: whole-batch Mpixels/s, the row engine per image, reference digests."""
    import numpy as np
    inputs = np.load(os.path.join(ROOT, "tests", "golden", "suite_inputs.npz"))
    names = sorted(inputs.files)
    want = {e["image"]: e for e in golden["suite"]}
    imgs = [inputs[n] for n in names]
    ctx = ctx_factory()
    stream = torch.cuda.current_stream().cuda_stream
    best = None
    for rep in range(3):
        dev = [torch.from_numpy(a.copy()).cuda() for a in imgs]
        filt = [torch.zeros(a.shape[0], dtype=torch.uint8, device="cuda") for a in imgs]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ctx.run([(d.data_ptr(), f.data_ptr(), a.shape[1], a.shape[0]) for d, f, a in zip(dev, filt, imgs)], STRENGTH, BLEED, stream=stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, ctx.engine_ms, dev, filt, res, [ctx.engine_info(i) for i in range(len(imgs))])
    dt, eng, dev, filt, res, infos = best
    px = sum(a.shape[0] * a.shape[1] for a in imgs)
    per = []
    for n, a, d, f, r, info in zip(names, imgs, dev, filt, res, infos):
        ok = r["status"] == 0 and "%016x" % P.fnv1a64(d.cpu().numpy(), P.SURVEY_FNV_BASIS) == want[n]["out"] and "%016x" % P.fnv1a64(f.cpu().numpy(), P.SURVEY_FNV_BASIS) == want[n]["filters"]
        per.append({"image": n, "size": [int(a.shape[1]), int(a.shape[0])], "bpp": r["bpp"], "engine": info["engine"], "attempts": info["attempts"], "digests_match_reference": bool(ok)})
    ctx.close()
    return {"workload": "BASELINE.json configs[2]: the reference's 11 suite images (2.94 Mpixels, 1/3/4 bytes per pixel) as one device-resident batch, s=19 b=2; best of 3",
            "value": round(px / dt / 1e6, 2), "unit": "Mpixels/s", "seconds": round(dt, 4), "engine_ms": round(eng, 2), "images": per,
            "all_digests_match_reference": all(p["digests_match_reference"] for p in per)}


def run_rank_shares(P, torch, ctx_factory, want):
    """What ONE rank of a node holds of BASELINE.json configs[3] at N = 8, 4, 2 -- the first 32, 64, 128 of the 256 frames -- as one device-resident
    batch on THIS GPU, best of two runs each; digests of every frame of the share the real reference has digests for.  The caller turns the rates
    into `projected_strong_scaling` = N x rate(256 / N frames) / rate(256 frames): what configs[3] can gain from N GPUs if nothing else gets in
    the way (the N-GPU run itself is the driver's)."""
    nmax = max(RANK_SHARES)
    base = [torch.from_numpy(P.synth_rgba(BATCH_W, BATCH_H, MODE, i)).cuda() for i in range(nmax)]
    ctx = ctx_factory()
    stream = torch.cuda.current_stream().cuda_stream
    out = []
    for n in RANK_SHARES:
        best = None
        for rep in range(2):
            dev = [b.clone() for b in base[:n]]
            filt = [torch.zeros(BATCH_H, dtype=torch.uint8, device="cuda") for _ in dev]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = ctx.run([(d.data_ptr(), f.data_ptr(), BATCH_W, BATCH_H) for d, f in zip(dev, filt)], STRENGTH, BLEED, stream=stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, ctx.engine_ms, dev, filt, res, ctx.engine_info(0), ctx.engine_info(n - 1))
        dt, eng, dev, filt, res, info0, info1 = best
        known = [i for i in range(n) if i in want]
        ok = all(r["status"] == 0 for r in res) and all(
            "%016x" % P.fnv1a64(dev[i].cpu().numpy(), P.SURVEY_FNV_BASIS) == want[i]["out"] and "%016x" % P.fnv1a64(filt[i].cpu().numpy(), P.SURVEY_FNV_BASIS) == want[i]["filters"] for i in known)
        out.append({"frames": n, "n_gpus_it_stands_for": BATCH_FRAMES // n, "value": round(n * BATCH_W * BATCH_H / dt / 1e6, 2), "seconds": round(dt, 4),
                    "engine_ms": round(eng, 2), "engine": info0["engine"] if info0["engine"] == info1["engine"] else "mixed", "attempts": info0["attempts"],
                    "digests_match_reference": bool(ok), "digests_checked_frames": known})
        del dev, filt
    ctx.close()
    del base
    torch.cuda.empty_cache()
    return out


SAT_FRAMES_PER_RANK, SAT_DISTINCT, SAT_BATCH = 512, 32, 512     # (SAT_BATCH: frames per call -- one call: the workgroups of 512 images are dispatched as CUs fall free,
                                                                  #  so a slow frame no longer holds a whole batch of 256: the frames differ by a third, 272 - 375 ms alone)


def run_batch_saturating(P, S, torch, ctx_factory, rank, world, barrier):
    """512 frames of 1920x1080 PER RANK (4096 in all at N = 8), in ONE device-resident batch (SAT_BATCH): a workload that keeps N GPUs
    busy, unlike configs[3] split N ways.  The frames are 32 distinct synthetic ones (indices 0, 1, 255 and 29 more), each
    uploaded once and cloned on the device; the digests of the copies of frames 0, 1 and 255 are checked."""
    idx = [0, 1, 255] + list(range(2, 2 + SAT_DISTINCT - 3))
    base = [torch.from_numpy(P.synth_rgba(BATCH_W, BATCH_H, MODE, i)).cuda() for i in idx]
    ctx = ctx_factory()
    total, eng, recs = 0.0, 0.0, []
    for part in range(SAT_FRAMES_PER_RANK // SAT_BATCH):
        dev = [base[k % SAT_DISTINCT].clone() for k in range(SAT_BATCH)]
        filt = [torch.zeros(BATCH_H, dtype=torch.uint8, device="cuda") for _ in dev]
        desc = [(d.data_ptr(), f.data_ptr(), BATCH_W, BATCH_H) for d, f in zip(dev, filt)]
        barrier()
        t0 = time.perf_counter()
        res = ctx.run(desc, STRENGTH, BLEED, stream=torch.cuda.current_stream().cuda_stream)
        barrier()
        total += time.perf_counter() - t0
        eng += ctx.engine_ms
        ok = all(r["status"] == 0 for r in res)
        for k in (0, 1, 2):
            recs.append(dict(frame=idx[k], ok=ok, out="%016x" % P.fnv1a64(dev[k].cpu().numpy(), P.SURVEY_FNV_BASIS),
                             filters="%016x" % P.fnv1a64(filt[k].cpu().numpy(), P.SURVEY_FNV_BASIS)))
        del dev, filt
    info = ctx.engine_info(0)
    ctx.close()
    return total, eng, recs, info


def run_read_side(P, torch, device):
    """SURVEY.md section 8 (f.2), the read side's inflate on the device (pngloss_hip_png_decode_batch_device_z, pl_inflate_core.h: one wave per zlib stream, three streams a CU):
    one 1280x720 RGBA file alone = MB/s of scanlines per stream; 768 such files in one call = the aggregate.  The files are made here (generator mode 0, PNG filter `sub` on every
    row, zlib level 6; 8 distinct ones, repeated); both times have the same call WITHOUT the inflate (scanlines up + inverse filters + expansion) taken off; the first and the last
    frame are read back and compared with the generator's pixels.  Informational: not the headline, not a baseline."""
    import ctypes as C
    import struct
    import zlib
    import numpy as np
    from pngloss_amd import lib as L
    W, H, DISTINCT, N = 1280, 720, 8, 768

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)

    frames = [P.synth_rgba(W, H, 0, i) for i in range(DISTINCT)]
    parsed = []
    for img in frames:
        rows = img.reshape(H, W * 4).astype(np.int16)
        sub = rows.copy(); sub[:, 4:] -= rows[:, :-4]
        raw = np.concatenate([np.full((H, 1), 1, np.uint8), (sub & 255).astype(np.uint8)], axis=1).tobytes()
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
        parsed.append(L.parse_png(png))
    lib = P.hip_lib()
    lib.pngloss_hip_png_decode_batch_device.argtypes = [C.c_void_p, C.POINTER(L.PngSource), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]
    lib.pngloss_hip_png_decode_batch_device_z.argtypes = [C.c_void_p, C.POINTER(L.PngZSource), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]
    hip = C.CDLL([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0])
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    ctx = P.HipContext(device)
    out = {"workload": f"the device inflate (read side, SURVEY 8 f.2): {W}x{H} RGBA PNG files (generator mode 0, filter sub, zlib level 6; {DISTINCT} distinct), one alone and {N} in one call, "
                       "file bytes -> RGBA8 frames in device memory; times are the call with the inflate minus the same call from inflated scanlines", "unit_per_stream": "MB/s of scanlines",
           "unit_aggregate": "GB/s of scanlines"}
    try:
        ok = True
        for n in (1, N):
            src = (L.PngSource * n)(); zsrc = (L.PngZSource * n)()
            for i in range(n):
                p = parsed[i % DISTINCT]
                src[i] = L.PngSource(p["scanlines"], W, H, 6, 8, None, 0, None, 0, None)
                zsrc[i] = L.PngZSource(p["zstream"], len(p["zstream"]), W, H, 6, 8, None, 0, None, 0)
            ptrs = (C.c_void_p * n)(); st = (C.c_int * n)()
            best_d = best_z = 1e9
            for rep in range(2):
                t0 = time.perf_counter(); rc1 = lib.pngloss_hip_png_decode_batch_device(ctx._ctx, src, n, ptrs, st, None)
                t1 = time.perf_counter(); rc2 = lib.pngloss_hip_png_decode_batch_device_z(ctx._ctx, zsrc, n, ptrs, st, None)
                t2 = time.perf_counter()
                ok = ok and rc1 == 0 and rc2 == 0 and not any(st[i] for i in range(n))
                best_d = min(best_d, t1 - t0); best_z = min(best_z, t2 - t1)
            for i in (0, n - 1):                       # (the frames of the last call: the one with the inflate)
                got = np.zeros(W * H * 4, np.uint8)
                ok = ok and hip.hipMemcpy(got.ctypes.data, ptrs[i], got.size, 2) == 0 and np.array_equal(got.reshape(H, W, 4), frames[i % DISTINCT])
            infl = max(1e-9, best_z - best_d)
            scan = H * (W * 4 + 1)
            if n == 1:
                out["per_stream"] = round(scan / 1e6 / infl, 2); out["one_file_ms"] = round(infl * 1e3, 1)
            else:
                out["aggregate"] = round(n * scan / 1e9 / infl, 2); out["files"] = n; out["call_ms"] = round(best_z * 1e3, 1); out["call_without_inflate_ms"] = round(best_d * 1e3, 1)
        out["frames_match_the_generator"] = bool(ok)
    finally:
        ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="skip the configs[3] batch leg and the saturating batch leg")
    ap.add_argument("--no-cpu-full-frame", action="store_true", help="cpu_baseline: only the 4096x1024 strip, not the full 4096x4096 frame (the full frame is ~18 s of ONE host core on an otherwise idle host)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the configs[4] sweep (8192x8192, 12 points) and the configs[2] suite batch (rank 0, N = 1 only)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import pngloss_amd as P
    from pngloss_amd import shard as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK in the environment) always go through RCCL, also for one rank, so that the
    # N=1 launch the driver does exercises exactly the code path of N=2,4,8
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    # ---- inputs resident in HBM before the timed region: one pristine copy per step (the path works in place) ----
    frame = P.synth_rgba(W, H, MODE, rank)
    src = torch.from_numpy(frame).cuda()
    nrun = args.warmup + args.steps
    work = [src.clone() for _ in range(nrun)]
    filt = [torch.zeros(H, dtype=torch.uint8, device="cuda") for _ in range(nrun)]
    ctx = P.HipContext(local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    engine_info = {}

    def step(i):
        res = ctx.run([(work[i].data_ptr(), filt[i].data_ptr(), W, H)], STRENGTH, BLEED, stream=stream)
        assert res[0]["status"] == 0 and res[0]["bpp"] == 4
        engine_info.update(ctx.engine_info(0))
        return ctx.engine_ms, ctx.total_ms

    # PCIe legs, reported separately and never part of `value` (inputs are resident before the timed region)
    pinned = torch.from_numpy(frame).pin_memory()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    scratch = torch.empty_like(src)
    ev[0].record(); scratch.copy_(pinned, non_blocking=True); ev[1].record(); pinned.copy_(scratch, non_blocking=True); ev[2].record()
    torch.cuda.synchronize()
    h2d_ms, d2h_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    del scratch

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    engine_ms = []
    step_wall_ms = []
    for i in range(args.warmup, nrun):
        ts = time.perf_counter()
        e, _ = step(i)
        step_wall_ms.append((time.perf_counter() - ts) * 1e3)
        engine_ms.append(e)
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())

    # ---- correctness of what was just timed: digests of the last step, gathered over ranks ----
    out_last = work[nrun - 1].cpu().numpy()
    filt_last = filt[nrun - 1].cpu().numpy()
    rec = [dict(index=rank, out="%016x" % P.fnv1a64(out_last, P.SURVEY_FNV_BASIS),
                filters="%016x" % P.fnv1a64(filt_last, P.SURVEY_FNV_BASIS), engine_ms=sum(engine_ms) / len(engine_ms))]
    try:
        records = S.gather_records(rec)
    except Exception as exc:          # never lose the measurement to a failed record gather
        print(f"bench.py: record gather failed on rank {rank}: {exc!r}", file=sys.stderr)
        records = rec if rank == 0 else rec

    # ---- cpu_baseline on the configuration the metric is quoted on: the reference on the FULL frame, one host core, ON AN OTHERWISE IDLE HOST: it runs here, between
    #      the timed headline steps and the GPU batch legs, and this process waits for it (round 5 ran it beside the batch legs, whose launch thread, clones and uploads
    #      kept the host busy: the ratio then depended on what else bench.py happened to be doing -- the advisor's finding) ----
    full_ff = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_cpu_full_frame:
        import multiprocessing as mp
        mpc = mp.get_context("spawn")       # (not fork: this process holds a HIP context)
        full_q = mpc.Queue()
        full_proc = mpc.Process(target=_full_frame_worker, args=(full_q,), daemon=True)
        torch.cuda.synchronize()
        full_proc.start()
        try:
            full_ff = full_q.get(timeout=240)
            full_proc.join(timeout=10)
        except Exception as exc:          # informational only
            full_ff = {"error": repr(exc)}

    # ---- the image-batch leg (BASELINE.json configs[3]); outside the timed region of `value` ----
    batch = None
    batch_sat = None
    shares = None
    if not args.no_batch:
        del work, filt
        torch.cuda.empty_cache()
        bdt, beng, brecs = run_batch(P, S, torch, lambda: batch_ctx(P, local_rank), rank, world, local_rank, barrier)
        tb = torch.tensor([bdt], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        try:
            allrecs = S.gather_records(brecs)
            engs = S.gather_records([dict(index=rank, engine_ms=beng, frames=len(brecs))])
        except Exception as exc:
            print(f"bench.py: batch record gather failed on rank {rank}: {exc!r}", file=sys.stderr)
            allrecs, engs = brecs, [dict(index=rank, engine_ms=beng, frames=len(brecs))]
        batch = (float(tb.item()), allrecs, engs)
        sdt, seng, srecs, sinfo = run_batch_saturating(P, S, torch, lambda: batch_ctx(P, local_rank), rank, world, barrier)
        ts = torch.tensor([sdt], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        try:
            sall = S.gather_records([dict(index=rank, engine_ms=seng, recs=srecs, engine=sinfo)])
        except Exception as exc:
            print(f"bench.py: saturating batch record gather failed on rank {rank}: {exc!r}", file=sys.stderr)
            sall = [dict(index=rank, engine_ms=seng, recs=srecs, engine=sinfo)]
        batch_sat = (float(ts.item()), sall)
        if rank == 0 and world == 1:
            want1080 = {k: v for k, v in BATCH_KNOWN.items() if v}
            try:
                shares = run_rank_shares(P, torch, lambda: batch_ctx(P, local_rank), want1080)
            except Exception as exc:          # never lose the headline to a side leg
                shares = {"error": repr(exc)}

    if rank == 0:
        golden = json.load(open(os.path.join(ROOT, "tests", "golden", "digests.json")))
        g = [e for e in golden["synthetic"] if e["width"] == W and e["height"] == H and e["strength"] == STRENGTH][0]
        bit_exact = records[0]["out"] == g["out"] and records[0]["filters"] == g["filters"]
        px = W * H
        value = world * px * args.steps / elapsed_max / 1e6
        eng_ms = sum(r["engine_ms"] for r in records) / len(records)
        achieved = ALGO_BYTES_PER_PIXEL * px / (eng_ms * 1e-3) / 1e9
        traffic = None          # HBM bytes per launch from the last committed rocprofv3 PMC passes (tools/pmc_to_json.py)
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC)))["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            pass
        line = {
            "metric": "Mpixels/s (filter+quantise path), 4096x4096 RGBA8 s=19; bit-exact vs ref",
            "value": round(value, 4), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 3), "step_ms_rank0": [round(v, 2) for v in step_wall_ms], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: one 4096x4096 synthetic RGBA8 frame per GPU "
                                   "(Appendix B generator mode 0, frame=rank), strength 19, bleed 2, "
                                   "row_filters requested; device-resident in, device-resident out",
                       "images_per_gpu": 1, "parallelism": f"image-parallel x{world}, no data-path collective"},
            "bit_exact_vs_reference_digest": bool(bit_exact),
            "roofline": {"bound": "hbm", "kernel": "row engine: " + str(engine_info.get("engine")) + (" (four launches per row attempt: seg_k_ctl [control of this attempt + validation of the attempt before, side by side], seg_k_enum, seg_k_chain, seg_k_replay)" if engine_info.get("engine") == "segment-parallel" else " (pl_engine)"),
                         "achieved": round(achieved, 6), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "engine_ms_per_launch": round(eng_ms, 3),
                         "engine": engine_info,
                         "engine_kernels": engine_kernels(),
                         "chain_bound": {"ns_per_pixel_step": round(eng_ms * 1e6 / px, 1),
                                         "us_per_row_attempt": round(eng_ms * 1e3 / max(1, engine_info.get("attempts", 0)), 2) if engine_info.get("engine") == "segment-parallel" else None,
                                         "note": "secondary, honest bound (SURVEY 8d): the rows of an image are strictly serial (the winner of row y seeds "
                                                 "row y+1), and so is a row's x-chain in the reference.  The segment-parallel engine cuts the x-chain into "
                                                 "32-pixel segments (state enumeration + map composition + exact validation, DESIGN.md section 4), so what is "
                                                 "left in series is H row attempts of four dependent launches each; the exact validation of an attempt runs next to the control kernel of the next one (the decision is optimistic, a failed validation voids the attempt under way)"},
                         "note": "bound by the row-to-row dependency chain (DESIGN.md), not by HBM; algorithmic bytes = 8 B/px * 16.78 Mpx = 134.2 MB "
                                 "per engine run, measured with HIP events around the engine's launches on the launch stream; traffic = "
                                 "FETCH_SIZE*2 + WRITE_SIZE of separate rocprofv3 --pmc passes (static: the committed profiles/" + PMC_TRAFFIC + ", not measured by this invocation)"},
            "bandwidth_kernels": bandwidth_kernels(),
        }
        # the static blocks above quote committed profiles: say what code they were taken at, and whether that is the code running now
        cur = P.source_digest()
        stamps = [profile_stamp(KERNEL_STATS), profile_stamp(PMC_TRAFFIC)]
        line["static_profiles"] = {"current_source_digest": cur, "files": stamps,
                                   "stale": [st["file"] for st in stamps if st["source_digest"] != cur],
                                   "note": "source_digest = sha256 over pngloss_amd/csrc/*.{hip,h,c} (pngloss_amd.source_digest); a file listed under `stale` was profiled at other "
                                           "kernel sources than the ones this line was measured with (round <= 4 profiles carry no stamp and always count as stale)"}
        line["static_profile_head"] = stamps[0]["head"]
        if line["static_profiles"]["stale"]:
            print("bench.py: WARNING: static profile blocks come from other sources than this tree: %s" % ", ".join(line["static_profiles"]["stale"]), file=sys.stderr)
        line["transfers"] = {"h2d_ms": round(h2d_ms, 3), "d2h_ms": round(d2h_ms, 3), "bytes_each_way": W * H * 4,
                             "note": "pinned 64 MiB frame over PCIe, outside the timed region; with both legs one step "
                                     "would take %.1f ms" % (elapsed_max / args.steps * 1e3 + h2d_ms + d2h_ms)}
        if world == 1 and not args.no_cpu_baseline:
            # the write side of the same frame, outside the timed region: scanline filtering + deflate on
            # the device; reported next to the headline, never part of `value`
            try:
                t_ws = time.perf_counter()
                _, _, streams = ctx.run_host_zlib([frame], STRENGTH, BLEED)
                ctype, zbytes, blocks = streams[0]
                line["write_side"] = {"gpu_deflate_ms": round(ctx.deflate_ms, 2), "zlib_stream_bytes": len(zbytes),
                                      "scanline_bytes": (W * {0: 1, 4: 2, 2: 3, 6: 4}[ctype] + 1) * H,
                                      "host_call_ms": round((time.perf_counter() - t_ws) * 1e3, 1),
                                      "note": "pngloss_hip_optimize_batch_host_zlib on the same frame from host memory: upload + "
                                              "hot path + scanline filtering + GPU deflate + download of the zlib stream"}
            except Exception as exc:          # informational only
                line["write_side"] = {"error": repr(exc)}
            line["cpu_baseline"] = cpu_baseline(frame)
            line["speedup_vs_cpu_baseline_strip"] = round(value / line["cpu_baseline"]["value"], 2)
            line["speedup_vs_cpu_baseline"] = line["speedup_vs_cpu_baseline_strip"]
            if full_ff is not None:
                try:
                    ff = full_ff
                    assert ff["rc"] == 0
                    fv = W * H / ff["seconds"] / 1e6
                    line["cpu_baseline"]["full_frame"] = {"value": round(fv, 4), "unit": "Mpixels/s", "cores": 1, "kind": ff["kind"], "seconds": round(ff["seconds"], 2),
                                                          "sample": f"the whole {W}x{H} frame the metric is quoted on, s={STRENGTH} b={BLEED}, one thread, host otherwise idle (between the timed steps and the batch legs of this invocation)",
                                                          "digests_match_reference": ff["out"] == g["out"] and ff["filters"] == g["filters"]}
                    line["speedup_vs_cpu_baseline"] = round(value / fv, 2)          # the like-for-like ratio: full frame against full frame
                except Exception as exc:          # informational only
                    line["cpu_baseline"]["full_frame"] = {"error": repr(exc)}
            line["speedup_vs_build_container_reference"] = round(value / BUILD_CONTAINER_REFERENCE_MPX, 2)
        if world == 1 and not args.no_sweep:
            for key, fn in (("suite_batch", run_suite_batch), ("sweep_8192", run_sweep_8192)):
                try:
                    line[key] = fn(P, torch, lambda: batch_ctx(P, local_rank), golden)
                except Exception as exc:          # never lose the headline to a side leg
                    line[key] = {"error": repr(exc)}
            try:
                line["read_side"] = run_read_side(P, torch, local_rank)
            except Exception as exc:
                line["read_side"] = {"error": repr(exc)}
        if batch is not None:
            bt, brecs, engs = batch
            want = {e["frame"]: e for e in golden["synthetic"] if (e["width"], e["height"]) == (BATCH_W, BATCH_H)}
            want.update({k: v for k, v in BATCH_KNOWN.items() if v})
            checked = [r for r in brecs if "out" in r]
            line["batch"] = {"workload": f"BASELINE.json configs[3]: {BATCH_FRAMES} synthetic {BATCH_W}x{BATCH_H} RGBA8 frames, s={STRENGTH} b={BLEED}, "
                                         f"contiguous split over {world} GPU(s), one device-resident batch per rank (strong scaling of a fixed batch)",
                             "value": round(BATCH_FRAMES * BATCH_W * BATCH_H / bt / 1e6, 2), "unit": "Mpixels/s", "seconds": round(bt, 4),
                             "frames_per_gpu": [e["frames"] for e in engs], "engine_ms_per_rank": [round(e["engine_ms"], 2) for e in engs],
                             "all_status_ok": all(r["status"] == 0 for r in brecs) and len(brecs) == BATCH_FRAMES,
                             "digests_match_reference": bool(checked) and all(r["out"] == want[r["index"]]["out"] and r["filters"] == want[r["index"]]["filters"] for r in checked),
                             "digests_checked_frames": [r["index"] for r in checked]}
            # the STRONG-scaling number of the fixed configs[3] batch, at top level for every N: a SCALE record plots `value` (weak scaling of one 4096^2 frame per
            # rank: linear by construction) -- this is the honest curve beside it
            line["batch_value"] = line["batch"]["value"]
            line["batch_unit"] = "Mpixels/s (BASELINE.json configs[3]: 256 frames of 1920x1080 split over n_gpus ranks; strong scaling)"
            line["launch_groups_option"] = 3
        if shares is not None and batch is not None:
            r256 = BATCH_FRAMES * BATCH_W * BATCH_H / batch[0] / 1e6
            if isinstance(shares, list):
                line["batch_rank_share"] = {"workload": f"the share ONE rank holds of BASELINE.json configs[3] at N = 8 / 4 / 2: the first 32 / 64 / 128 of its {BATCH_FRAMES} frames ({BATCH_W}x{BATCH_H}, s={STRENGTH} b={BLEED}) "
                                                        "as one device-resident batch on this GPU, best of 2", "unit": "Mpixels/s", "shares": shares, "rate_256_frames_one_gpu": round(r256, 2),
                                            "projected_strong_scaling": {str(BATCH_FRAMES // sh["frames"]): round(BATCH_FRAMES // sh["frames"] * sh["value"] / r256, 2) for sh in shares},
                                            "all_digests_match_reference": all(sh["digests_match_reference"] for sh in shares),
                                            "note": "projected_strong_scaling[N] = N x rate(256/N frames on one GPU) / rate(256 frames on one GPU): the speed-up configs[3] can get from N GPUs "
                                                    "(image-level sharding, no data-path collective), measured on ONE GPU; the N-GPU run is the driver's"}
            else:
                line["batch_rank_share"] = shares
        if batch_sat is not None:
            st, sall = batch_sat
            want = {e["frame"]: e for e in golden["synthetic"] if (e["width"], e["height"]) == (BATCH_W, BATCH_H)}
            chk = [r for e in sall for r in e["recs"]]
            nfr = SAT_FRAMES_PER_RANK * world
            line["batch_saturating"] = {"workload": f"{SAT_FRAMES_PER_RANK} synthetic {BATCH_W}x{BATCH_H} RGBA8 frames PER GPU ({nfr} in all), s={STRENGTH} b={BLEED}, "
                                                    f"device-resident batches of {SAT_BATCH}; {SAT_DISTINCT} distinct frames cloned on the device",
                                        "value": round(nfr * BATCH_W * BATCH_H / st / 1e6, 2), "unit": "Mpixels/s", "seconds": round(st, 4), "scaling": "weak",
                                        "engine_ms_per_rank": [round(e["engine_ms"], 2) for e in sall],
                                        "engine": sall[0].get("engine", {}).get("engine"),
                                        "digests_match_reference": bool(chk) and all(r["ok"] and r["out"] == want[r["frame"]]["out"] and r["filters"] == want[r["frame"]]["filters"] for r in chk),
                                        "note": "configs[3] (256 frames in all) is a fixed batch: what N GPUs can make of it is batch_rank_share's projected_strong_scaling (a rank's share runs on the segment engine); "
                                                "this leg is the saturated GPU instead: every GPU gets 512 frames in one call, two per CU, dispatched as CUs fall free (a frame takes 272 - 375 ms alone, so a call of 256 waits for its slowest)"}
        print(json.dumps(line), flush=True)
    ctx.close()
    if use_dist:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
