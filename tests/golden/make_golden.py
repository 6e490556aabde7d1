"""Generate the committed golden fixtures from the REAL reference (oracle/_ref/libpngloss_ref.so).

Run in the build container only (needs /root/reference for the suite PNGs and oracle/_ref built by
`make -C oracle`):   python tests/golden/make_golden.py

Outputs (data only -- inputs and expected outputs, no reference source in any form):
  tests/golden/synth_cases.npz   expected RGBA8 output + filter IDs of optimize_with_rows for seeded synthetic inputs
                                 (inputs are regenerated from pngloss_amd.synth_rgba, so only outputs are stored)
  tests/golden/suite_small.npz   rose.png / david.png / tux.png of the reference's suite, decoded to RGBA8 (the same
                                 bytes rwpng_read_image24 yields), plus the expected outputs
  tests/golden/suite_inputs.npz  ALL eleven suite images decoded to RGBA8 (inputs only; their expected outputs are the
                                 reference-measured digests in digests.json) -- BASELINE.json configs[2] as one batch
  tests/golden/digests.json      FNV-1a-64 digests (SURVEY.md Appendix B basis) of inputs/outputs/filters for every
                                 BASELINE.json configuration, re-measured here against the real reference for all
                                 sizes up to 1920x1080, and carried over from SURVEY.md Appendix B (measured by the
                                 survey with the same reference build) for 4096x4096 and 8192x8192
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pngloss_amd as P  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpngloss_ref.so"))
ref.optimize_with_rows.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_void_p, C.c_bool, C.c_uint8, C.c_long]
ref.optimize_with_rows.restype = C.c_int


def run_ref(img, s, b, filters=True):
    h, w, _ = img.shape
    out = np.ascontiguousarray(img).copy()
    f = np.zeros(h, np.uint8)
    rows = (C.c_void_p * h)(*[out.ctypes.data + y * w * 4 for y in range(h)])
    assert ref.optimize_with_rows(rows, w, h, f.ctypes.data if filters else None, False, s, b) == 0
    return out, f


#: (width, height, mode, strength, bleed, frame, want_filters)
SYNTH_CASES = (
    [(64, 48, m, 19, 2, 0, True) for m in range(6)]
    + [(64, 48, 0, s, b, 0, True) for (s, b) in [(0, 2), (20, 1), (40, 2), (85, 8), (255, 1), (19, 32767)]]
    + [(64, 48, m, s, b, 0, True) for m in (1, 5) for (s, b) in [(40, 2), (85, 8)]]
    + [(w, h, 1, 19, 2, 0, True) for (w, h) in [(1, 1), (2, 3), (5, 1), (1, 7)]]
    + [(96, 64, m, 19, 2, 3, False) for m in (0, 3, 4, 5)]      # row_filters == NULL mode: every row adaptive
    + [(130, 9, m, 19, 2, 1, True) for m in (0, 2, 3, 4)]        # ragged width (two 64-pixel chunks + 2)
)


def case_key(c):
    return "w%d_h%d_m%d_s%d_b%d_f%d_%s" % (c[0], c[1], c[2], c[3], c[4], c[5], "ids" if c[6] else "null")


def main():
    arrays = {}
    for c in SYNTH_CASES:
        w, h, m, s, b, fr, filt = c
        out, f = run_ref(P.synth_rgba(w, h, m, fr), s, b, filt)
        arrays[case_key(c) + "/out"] = out
        if filt:
            arrays[case_key(c) + "/filters"] = f
    np.savez_compressed(os.path.join(HERE, "synth_cases.npz"), **arrays)

    from PIL import Image
    suite = {}
    suite_inputs = {}
    digests = {"basis": "0x%016x" % P.SURVEY_FNV_BASIS, "synthetic": [], "suite": []}
    for name in sorted(os.listdir("/root/reference/suite")):
        if not name.endswith(".png"):
            continue
        img = np.array(Image.open(os.path.join("/root/reference/suite", name)).convert("RGBA"))
        suite_inputs[name[:-4]] = img
        out, f = run_ref(img, 19, 2)
        digests["suite"].append(dict(image=name[:-4], width=img.shape[1], height=img.shape[0], strength=19, bleed=2,
                                     **{"in": "%016x" % P.fnv1a64(img, P.SURVEY_FNV_BASIS)},
                                     out="%016x" % P.fnv1a64(out, P.SURVEY_FNV_BASIS),
                                     filters="%016x" % P.fnv1a64(f, P.SURVEY_FNV_BASIS)))
        if name[:-4] in ("rose", "david", "tux"):
            suite[name[:-4] + "/in"] = img
            suite[name[:-4] + "/out"] = out
            suite[name[:-4] + "/filters"] = f
    np.savez_compressed(os.path.join(HERE, "suite_small.npz"), **suite)
    np.savez_compressed(os.path.join(HERE, "suite_inputs.npz"), **suite_inputs)

    measured = [(64, 48, m, 19, 2, 0) for m in range(6)] + [(512, 512, 0, 19, 2, 0), (512, 512, 1, 19, 2, 0),
                (1920, 1080, 0, 19, 2, 0), (1920, 1080, 0, 19, 2, 1), (1920, 1080, 0, 19, 2, 255)]
    for (w, h, m, s, b, fr) in measured:
        img = P.synth_rgba(w, h, m, fr)
        out, f = run_ref(img, s, b)
        digests["synthetic"].append(dict(width=w, height=h, mode=m, strength=s, bleed=b, frame=fr, source="measured here",
                                         **{"in": "%016x" % P.fnv1a64(img, P.SURVEY_FNV_BASIS)},
                                         out="%016x" % P.fnv1a64(out, P.SURVEY_FNV_BASIS),
                                         filters="%016x" % P.fnv1a64(f, P.SURVEY_FNV_BASIS)))
    # too slow to re-measure on every regeneration (33 s .. 3.5 min each on the reference): SURVEY.md Appendix B
    carried = [
        (4096, 4096, 0, 19, 2, 0, "3d9e6d9e8520eff1", "81f4506aed842e71", "8742102128583203"),
        (8192, 8192, 0, 0, 2, 0, "b4752ef2264ad332", "b4752ef2264ad332", "251040a11bbe77a3"),
        (8192, 8192, 0, 20, 1, 0, "b4752ef2264ad332", "9c9e7c6b30f625fb", "7e5a70dff8b685d3"),
        (8192, 8192, 0, 20, 2, 0, "b4752ef2264ad332", "50b5190b7138def5", "20543521136c3023"),
        (8192, 8192, 0, 20, 8, 0, "b4752ef2264ad332", "d2df8e48e5bb056d", "4396f2bdbb8eb0e3"),
        (8192, 8192, 0, 40, 1, 0, "b4752ef2264ad332", "c708b38af211836c", "40bffc8220a064b3"),
        (8192, 8192, 0, 40, 2, 0, "b4752ef2264ad332", "c04a72b425c63cb8", "2e2457b3eaae7713"),
        (8192, 8192, 0, 40, 8, 0, "b4752ef2264ad332", "6282d9c9fa28056c", "9cbd5184865615f3"),
        (8192, 8192, 0, 85, 1, 0, "b4752ef2264ad332", "73662410d67fbec3", "c42811ba9b2420d3"),
        (8192, 8192, 0, 85, 2, 0, "b4752ef2264ad332", "ac67969c1e2df38d", "951fbb3b22447c83"),
        (8192, 8192, 0, 85, 8, 0, "b4752ef2264ad332", "b8d6b9ac26832124", "40d35c4ddc6f53a3"),
    ]
    for (w, h, m, s, b, fr, di, do, df) in carried:
        digests["synthetic"].append(dict(width=w, height=h, mode=m, strength=s, bleed=b, frame=fr,
                                         source="SURVEY.md Appendix B (reference run by the survey)", **{"in": di}, out=do, filters=df))
    with open(os.path.join(HERE, "digests.json"), "w") as fh:
        json.dump(digests, fh, indent=1)
    print("wrote", len(SYNTH_CASES), "synthetic cases,", len(suite) // 3, "suite images,", len(digests["synthetic"]) + len(digests["suite"]), "digests")


if __name__ == "__main__":
    main()
