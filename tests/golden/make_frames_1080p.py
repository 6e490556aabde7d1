"""More reference digests for BASELINE.json configs[3] (256 frames of 1920x1080, generator mode 0, s=19 b=2): the REAL reference
(oracle/_ref/libpngloss_ref.so, build container only) on EVERY frame index 0..255 (round 6; a spread of 20 before), one process per frame,
~2.5 minutes on 8 cores.
   python tests/golden/make_frames_1080p.py      ->  tests/golden/digests_1080p.json   (data only: digests of inputs / outputs / filters)
digests.json holds frames 0, 1 and 255 as the survey measured them; this file holds all 256 (those three included, and they
must agree): the 256-frame batch test and the batch legs of bench.py check every frame they run."""
import ctypes as C
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pngloss_amd as P  # noqa: E402

FRAMES = list(range(256))
W, H, S, B = 1920, 1080, 19, 2


def one(frame):
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpngloss_ref.so"))
    ref.optimize_with_rows.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_void_p, C.c_bool, C.c_uint8, C.c_long]
    ref.optimize_with_rows.restype = C.c_int
    img = P.synth_rgba(W, H, 0, frame)
    out = img.copy()
    f = np.zeros(H, np.uint8)
    rows = (C.c_void_p * H)(*[out.ctypes.data + y * W * 4 for y in range(H)])
    assert ref.optimize_with_rows(rows, W, H, f.ctypes.data, False, S, B) == 0
    return dict(width=W, height=H, mode=0, strength=S, bleed=B, frame=frame,
                **{"in": "%016x" % P.fnv1a64(img, P.SURVEY_FNV_BASIS)}, out="%016x" % P.fnv1a64(out, P.SURVEY_FNV_BASIS),
                filters="%016x" % P.fnv1a64(f, P.SURVEY_FNV_BASIS))


if __name__ == "__main__":
    with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
        recs = pool.map(one, FRAMES, chunksize=1)
    json.dump({"basis": "0x%016x" % P.SURVEY_FNV_BASIS, "generator": "tests/golden/make_frames_1080p.py (real reference, oracle/_ref)", "frames": recs},
              open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "digests_1080p.json"), "w"), indent=1)
    print(len(recs), "frames")
