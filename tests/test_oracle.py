"""CPU tests: pin the oracle (oracle/pngloss_port.c) to the reference.

Sources of truth, in order of strength:
  1. the REAL reference compiled from /root/reference (oracle/_ref) -- only where it was built (the build container)
  2. tests/golden/*.npz -- outputs of that same reference build, committed as data
  3. tests/golden/digests.json -- digests of the BASELINE.json configurations
"""
import ctypes as C

import numpy as np
import pytest

import pngloss_amd as P
from tests import util as U


@pytest.mark.parametrize("case", U.SYNTH_CASES, ids=U.case_key)
@pytest.mark.parametrize("variant", [0, 1, 2], ids=["plain", "gpu-shaped", "band-leader"])
def test_port_matches_golden_synthetic(case, variant):
    w, h, m, s, b, fr, filt = case
    g = U.load_npz("synth_cases.npz")
    out, f = U.run_port(P.synth_rgba(w, h, m, fr), s, b, filt, variant=variant)
    assert np.array_equal(out, g[U.case_key(case) + "/out"])
    if filt:
        assert np.array_equal(f, g[U.case_key(case) + "/filters"])


@pytest.mark.parametrize("name", ["rose", "david", "tux"])
def test_port_matches_golden_suite_images(name):
    """rose = RGB (3 B/px), david = gray (1 B/px, BASELINE.json configs[0]), tux = palette+tRNS -> RGBA with alpha 0."""
    g = U.load_npz("suite_small.npz")
    out, f = U.run_port(g[name + "/in"], 19, 2)
    assert np.array_equal(out, g[name + "/out"])
    assert np.array_equal(f, g[name + "/filters"])


def test_port_matches_reference_digests_up_to_512():
    for e in U.load_digests()["synthetic"]:
        if e["width"] * e["height"] > 512 * 512:
            continue
        img = P.synth_rgba(e["width"], e["height"], e["mode"], e["frame"])
        assert "%016x" % P.fnv1a64(img, P.SURVEY_FNV_BASIS) == e["in"]
        out, f = U.run_port(img, e["strength"], e["bleed"])
        assert "%016x" % P.fnv1a64(out, P.SURVEY_FNV_BASIS) == e["out"], e
        assert "%016x" % P.fnv1a64(f, P.SURVEY_FNV_BASIS) == e["filters"], e


def test_port_matches_reference_digest_1080p_frame():
    e = [e for e in U.load_digests()["synthetic"] if e["width"] == 1920 and e["frame"] == 255][0]
    out, f = U.run_port(P.synth_rgba(1920, 1080, 0, 255), 19, 2)
    assert "%016x" % P.fnv1a64(out, P.SURVEY_FNV_BASIS) == e["out"]
    assert "%016x" % P.fnv1a64(f, P.SURVEY_FNV_BASIS) == e["filters"]


@pytest.mark.skipif(U.ref() is None, reason="oracle/_ref not built (no /root/reference on this box)")
@pytest.mark.parametrize("variant", [0, 1, 2], ids=["plain", "gpu-shaped", "band-leader"])
def test_port_matches_real_reference_on_seeded_inputs(variant):
    for img, s, b, filt in U.seeded_cases(seed=11, n=36):
        o1, f1 = U.run_ref(img, s, b, filt)
        o2, f2 = U.run_port(img, s, b, filt, variant=variant)
        assert np.array_equal(o1, o2), (img.shape, s, b, filt)
        if filt:
            assert np.array_equal(f1, f2), (img.shape, s, b, filt)


@pytest.mark.skipif(U.ref() is None, reason="oracle/_ref not built (no /root/reference on this box)")
def test_golden_fixtures_are_what_the_reference_produces():
    g = U.load_npz("synth_cases.npz")
    for case in U.SYNTH_CASES[:8]:
        w, h, m, s, b, fr, filt = case
        out, f = U.run_ref(P.synth_rgba(w, h, m, fr), s, b, filt)
        assert np.array_equal(out, g[U.case_key(case) + "/out"])


def test_strength_zero_is_identity_for_every_class():
    """SURVEY.md section 4, property 1: s=0 leaves the pixels alone (filters are still chosen)."""
    for mode in range(6):
        img = P.synth_rgba(40, 12, mode, 2)
        out, f = U.run_port(img, 0, 2)
        assert np.array_equal(out, img)
        assert set(np.unique(f)) <= set(P.PNG_FILTER_FLAGS)


def test_row0_strength_retry_fires():
    """SURVEY.md section 4, property 4: with row_filters the first row must equal libpng's heuristic, which needs
    the strength-decrement retry of pngloss_image.c:266-274 on smooth gray input."""
    img = P.synth_rgba(96, 64, 4, 0)
    h, w = img.shape[:2]
    packed = np.ascontiguousarray(img[..., 1:2])
    lib = U.port()
    out = packed.copy()
    f = np.zeros(h, np.uint8)
    su = np.zeros(h, np.uint8)
    tr = U.PortTrace(None, su.ctypes.data, None, None)
    assert lib.port_optimize_packed(out.ctypes.data, w, h, 1, f.ctypes.data, 19, 2, C.byref(tr)) == 0
    assert su[0] < 19 and (su[1:] == 19).all()


# ---- building blocks --------------------------------------------------------------------------------------

def _predict(f, above, diag, left):
    if f == 0:
        return 0
    if f == 1:
        return left
    if f == 2:
        return above
    if f == 3:
        return (above + left) // 2
    p, pd = above - diag, left - diag
    pl, pa, pg = abs(p), abs(pd), abs(p + pd)
    return left if (pl <= pa and pl <= pg) else (above if pa <= pg else diag)


def test_orig_histograms_against_bruteforce():
    rng = np.random.default_rng(3)
    for bpp in (1, 2, 3, 4):
        h, w = 7, 9
        pix = rng.integers(0, 256, (h, w, bpp), dtype=np.uint8)
        got = np.zeros((5, 256), np.uint32)
        U.port().port_orig_histograms(pix.ctypes.data, w, h, bpp, got.ctypes.data)
        want = np.zeros((5, 256), np.uint32)
        for y in range(h):
            for x in range(w):
                for c in range(bpp):
                    left = int(pix[y, x - 1, c]) if x else 0
                    above = int(pix[y - 1, x, c]) if y else 0
                    diag = int(pix[y - 1, x - 1, c]) if (x and y) else 0
                    for f in range(5):
                        want[f, (int(pix[y, x, c]) - _predict(f, above, diag, left)) & 255] += 1
        assert np.array_equal(got, want)
        assert (got.sum(axis=1) == h * w * bpp).all()


def test_symbol_cost_is_bit_length_of_uintmax_div_freq():
    """optimize_state.c:338,565-572: ulog2(UINTMAX_MAX / f)."""
    rng = np.random.default_rng(5)
    freqs = list(range(1, 5000)) + [int(v) for v in rng.integers(1, 2**32, 3000)] + [2**k for k in range(32)] + [2**32 - 1]
    for f in freqs:
        assert U.port().port_symbol_cost(f) == ((2**64 - 1) // f).bit_length()
    assert U.port().port_symbol_cost(0) == 0


def test_sierra_split_conserves_the_error_and_matches_c_division():
    parts = (C.c_int * 5)()
    for bleed in (1, 2, 3, 8, 100, 32767):
        for d in list(range(-700, 701)) + [-32768, 32767, -20000, 20000]:
            U.port().port_sierra_split(d, bleed, C.byref(parts))
            t, h, f, v, rem = list(parts)
            d0 = int(d / bleed)  # C truncation
            assert 4 * t + 2 * h + 2 * f + v + rem == d0
            assert t == int(d0 / 16)


def test_adaptive_filter_against_bruteforce():
    rng = np.random.default_rng(9)
    for bpp in (1, 2, 3, 4):
        for trial in range(6):
            w = int(rng.integers(1, 30))
            row = rng.integers(0, 256, (w * bpp,), dtype=np.uint8)
            above = rng.integers(0, 256, (w * bpp,), dtype=np.uint8) if trial % 2 else None
            sums = [0] * 5
            for i in range(w * bpp):
                left = int(row[i - bpp]) if i >= bpp else 0
                ab = int(above[i]) if above is not None else 0
                dg = int(above[i - bpp]) if (above is not None and i >= bpp) else 0
                for f in range(5):
                    b = (int(row[i]) - _predict(f, ab, dg, left)) & 255
                    sums[f] += b if b < 128 else 256 - b
            want = sums.index(min(sums))
            got = U.port().port_adaptive_filter(above.ctypes.data if above is not None else None, row.ctypes.data, w, bpp)
            assert got == want


def test_restatement_is_clean_under_asan_and_ubsan(tmp_path):
    """the checker every GPU test trusts (oracle/pngloss_port.c, all three chain variants) built with
    -fsanitize=address,undefined must run without a report, agree between its variants, and print the digests of the ordinary build."""
    import os
    import subprocess
    src = [os.path.join(U.ROOT, "tests", "c", "port_sanitize_main.c"), os.path.join(U.ROOT, "pngloss_amd", "csrc", "pngloss_synth.c")]
    outs = []
    for tag, flags in (("plain", ["-O1"]), ("san", ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"])):
        exe = str(tmp_path / ("port_" + tag))
        subprocess.run(["gcc", "-std=gnu11", "-Wno-unknown-pragmas", "-Wno-unused-function"] + flags + ["-o", exe] + src, check=True, capture_output=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-2000:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] and outs[0].count("\n") == 15
