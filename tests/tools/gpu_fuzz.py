"""Randomised parity campaign on the GPU box: HIP path (C ABI) vs the CPU oracle over random shapes, contents, strengths
and bleed dividers, both row_filters modes, plus the device batch API with mixed images.
usage: gpu_fuzz.py [seconds] [seed] [big]     (big: shapes up to 1500 x 120, so that histogram counts and error rows grow)
FUZZ_ENGINES=seg,mix,  pins the row engine case by case (segment-parallel, alternating chain kinds, the library's choice); FUZZ_STRENGTH=s pins the strength"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pngloss_amd as P
from tests import util as U

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"


def make(rng):
    return U.fuzz_case(rng, BIG, int(os.environ["FUZZ_STRENGTH"]) if os.environ.get("FUZZ_STRENGTH") else None)    # (FUZZ_STRENGTH: a campaign at one strength: 0 = the row-statistics engine)


t0 = time.time(); n = 0; bad = 0
ctx = P.HipContext()
import torch
# FUZZ_WATCHDOG=seconds: a case that takes longer than that gets its number and every thread's Python stack printed, and the process ends (a hang then names its case and the call it sits in)
WATCHDOG = float(os.environ.get("FUZZ_WATCHDOG", "0"))
if WATCHDOG:
    import faulthandler
while time.time() - t0 < budget:
    if WATCHDOG:
        faulthandler.cancel_dump_traceback_later()
        with open("gpurun_out/fuzz_current_case.txt", "w") as fh: fh.write(f"{n} seed {sys.argv[2] if len(sys.argv) > 2 else 1234} t={time.time() - t0:.1f}\n")
        faulthandler.dump_traceback_later(WATCHDOG, exit=True)
    # the row engine is pinned case by case (the variable is read per call): by default every other case the band-leader chains (no
    # adaptive fallback to the round-1 chains on slow rows); FUZZ_ENGINES="seg,mix," cycles through the named ones ("" = the library's choice)
    engines = os.environ.get("FUZZ_ENGINES", ",lead").split(",")
    eng = engines[n % len(engines)]
    if eng: os.environ["PNGLOSS_HIP_ENGINE"] = eng
    else: os.environ.pop("PNGLOSS_HIP_ENGINE", None)
    if n % 10 == 9:     # device batch of 5 mixed images
        items = [make(rng) for _ in range(5)]
        s, b = items[0][1], items[0][2]
        dev = [torch.from_numpy(it[0].copy()).cuda() for it in items]
        flt = [torch.zeros(it[0].shape[0], dtype=torch.uint8, device="cuda") for it in items]
        ctx.run([(d.data_ptr(), f.data_ptr(), it[0].shape[1], it[0].shape[0]) for d, f, it in zip(dev, flt, items)], s, b)
        for d, f, it in zip(dev, flt, items):
            o1, f1 = U.run_port(it[0], s, b)
            if not (np.array_equal(o1, d.cpu().numpy()) and np.array_equal(f1, f.cpu().numpy())):
                bad += 1; print("BATCH MISMATCH", it[0].shape, s, b, flush=True)
    else:
        img, s, b, filt = make(rng)
        o1, f1 = U.run_port(img, s, b, filt)
        try:
            o2, f2 = P.optimize_with_rows(img, s, b, want_filters=filt)
        except RuntimeError as exc:       # the library refused or failed: say which case, keep it, go on
            bad += 1
            np.save(f"gpurun_out/fuzz_error_{n}.npy", img)
            print("ERROR", n, img.shape, s, b, filt, "engine", repr(eng), exc, flush=True)
            n += 1
            continue
        if not (np.array_equal(o1, o2) and (not filt or np.array_equal(f1, f2))):
            bad += 1
            np.save(f"gpurun_out/fuzz_fail_{n}.npy", img)
            print("MISMATCH", n, img.shape, s, b, filt, int((o1 != o2).sum()), flush=True)
    n += 1
if WATCHDOG: faulthandler.cancel_dump_traceback_later()
print(f"fuzz: {n} cases in {time.time() - t0:.0f} s, {bad} mismatches")
sys.exit(1 if bad else 0)
