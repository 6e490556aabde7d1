"""Shared helpers for the tests: ctypes access to the CPU oracle (tests may use oracle/, the product may not)."""
import ctypes as C
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

_ROWS_SIG = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_void_p, C.c_bool, C.c_uint8, C.c_long]


class PortTrace(C.Structure):
    _fields_ = [("cost", C.c_void_p), ("strength_used", C.c_void_p), ("winner", C.c_void_p), ("final_hist", C.c_void_p)]


_port = None
_ref = None


def port():
    global _port
    if _port is None:
        lib = C.CDLL(os.path.join(ROOT, "oracle", "libpngloss_port.so"))
        lib.port_optimize_with_rows.argtypes = _ROWS_SIG
        lib.port_optimize_with_rows.restype = C.c_int
        lib.port_optimize_packed.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint, C.c_long, C.POINTER(PortTrace)]
        lib.port_optimize_packed.restype = C.c_int
        lib.port_orig_histograms.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.port_adaptive_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.port_adaptive_filter.restype = C.c_int
        lib.port_sierra_split.argtypes = [C.c_int, C.c_long, C.POINTER(C.c_int * 5)]
        lib.port_symbol_cost.argtypes = [C.c_uint32]
        lib.port_symbol_cost.restype = C.c_uint
        lib.port_set_chain_variant.argtypes = [C.c_int]
        _port = lib
    return _port


def ref():
    """The REAL reference hot path (oracle/_ref, built from /root/reference by oracle/Makefile) or None."""
    global _ref
    if _ref is None:
        path = os.path.join(ROOT, "oracle", "_ref", "libpngloss_ref.so")
        if not os.path.exists(path):
            return None
        lib = C.CDLL(path)
        lib.optimize_with_rows.argtypes = _ROWS_SIG
        lib.optimize_with_rows.restype = C.c_int
        _ref = lib
    return _ref


def _run_rows(fn, img, s, b, filters=True):
    img = np.ascontiguousarray(img)
    h, w, _ = img.shape
    out = img.copy()
    f = np.zeros(h, np.uint8)
    rows = (C.c_void_p * max(h, 1))(*[out.ctypes.data + y * w * 4 for y in range(h)])
    rc = fn(rows, w, h, f.ctypes.data if filters else None, False, s, b)
    assert rc == 0
    return out, (f if filters else None)


def run_port(img, s=19, b=2, filters=True, variant=0):
    port().port_set_chain_variant(variant)
    try:
        return _run_rows(port().port_optimize_with_rows, img, s, b, filters)
    finally:
        port().port_set_chain_variant(0)


def run_ref(img, s=19, b=2, filters=True):
    return _run_rows(ref().optimize_with_rows, img, s, b, filters)


def run_port_packed(packed, s=19, b=2, filters=True, trace=False):
    """port_optimize_packed on an (H, W, bpp) array; returns out, filters, (final_hist if trace)."""
    out = np.ascontiguousarray(packed).copy()
    h, w, bpp = out.shape
    f = np.zeros(h, np.uint8)
    hist = np.zeros(256, np.uint32)
    tr = PortTrace(None, None, None, hist.ctypes.data)
    rc = port().port_optimize_packed(out.ctypes.data, w, h, bpp, f.ctypes.data if filters else None, s, b, C.byref(tr) if trace else None)
    assert rc == 0
    return (out, (f if filters else None), hist) if trace else (out, (f if filters else None))


# (width, height, mode, strength, bleed, frame, want_filters) -- must match tests/golden/make_golden.py:SYNTH_CASES
SYNTH_CASES = (
    [(64, 48, m, 19, 2, 0, True) for m in range(6)]
    + [(64, 48, 0, s, b, 0, True) for (s, b) in [(0, 2), (20, 1), (40, 2), (85, 8), (255, 1), (19, 32767)]]
    + [(64, 48, m, s, b, 0, True) for m in (1, 5) for (s, b) in [(40, 2), (85, 8)]]
    + [(w, h, 1, 19, 2, 0, True) for (w, h) in [(1, 1), (2, 3), (5, 1), (1, 7)]]
    + [(96, 64, m, 19, 2, 3, False) for m in (0, 3, 4, 5)]
    + [(130, 9, m, 19, 2, 1, True) for m in (0, 2, 3, 4)]
)


def case_key(c):
    return "w%d_h%d_m%d_s%d_b%d_f%d_%s" % (c[0], c[1], c[2], c[3], c[4], c[5], "ids" if c[6] else "null")


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def load_digests():
    with open(os.path.join(GOLDEN, "digests.json")) as fh:
        return json.load(fh)


def load_digests_1080p():
    """frame -> reference digests of BASELINE.json configs[3]'s 1920x1080 frames: 0, 1, 255 from digests.json and all 256 of
    digests_1080p.json (tests/golden/make_frames_1080p.py: the real reference in the build container; the three frames both files hold must agree)."""
    want = {e["frame"]: e for e in load_digests()["synthetic"] if (e["width"], e["height"]) == (1920, 1080)}
    with open(os.path.join(GOLDEN, "digests_1080p.json")) as fh:
        for e in json.load(fh)["frames"]:
            if e["frame"] in want:
                assert (want[e["frame"]]["out"], want[e["frame"]]["filters"]) == (e["out"], e["filters"]), e["frame"]
            want[e["frame"]] = e
    return want


def seeded_cases(seed=7, n=24):
    """Random RGBA images covering all four byte-per-pixel classes, transparency, extreme strengths and bleeds."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        w = int(rng.integers(1, 150))
        h = int(rng.integers(1, 24))
        s = int(rng.choice([0, 1, 7, 15, 16, 19, 31, 32, 40, 85, 200, 255]))
        b = int(rng.choice([1, 2, 3, 8, 100, 32767]))
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        kind = i % 6
        if kind == 1:
            img[..., 3] = 255
        elif kind == 2:
            img[..., 0] = img[..., 1]; img[..., 2] = img[..., 1]
        elif kind == 3:
            img[..., 0] = img[..., 1]; img[..., 2] = img[..., 1]; img[..., 3] = 255
        elif kind == 4:
            img[..., 3] = np.where(rng.random((h, w)) < 0.3, 0, img[..., 3])
        elif kind == 5:
            img = (img // 86 * 127).astype(np.uint8)   # few distinct values: lots of saturated 0/254 runs and ties
        out.append((img, s, b, bool(i % 3)))
    return out


def png_scanlines_reference(rgba_out, filter_flags):
    """What a PNG encoder must deflate for the optimised image: (color_type, filter type per row, filtered rows).
    numpy restatement of the writer side of the reference (rwpng.c:558-609 colour type + gray repack, :477-501
    per-row filters, row 0 -- or every row when filter_flags is None -- by libpng's minimum-sum heuristic)."""
    img = np.asarray(rgba_out)
    h, w = img.shape[:2]
    gray = bool((img[..., 0] == img[..., 1]).all() and (img[..., 1] == img[..., 2]).all())
    opaque = bool((img[..., 3] == 255).all())
    if gray:
        raw = img[..., 1:2] if opaque else img[..., [1, 3]]
        ctype = 0 if opaque else 4
    else:
        raw = img[..., :3] if opaque else img
        ctype = 2 if opaque else 6
    ch = raw.shape[2]
    raw = raw.reshape(h, w * ch).astype(np.int32)
    zeros = np.zeros(w * ch, np.int32)
    ids = np.zeros(h, np.uint8)
    rows = np.zeros((h, w * ch), np.uint8)
    flag_to_id = {0x08: 0, 0x10: 1, 0x20: 2, 0x40: 3, 0x80: 4}
    for y in range(h):
        cur = raw[y]
        up = raw[y - 1] if y else zeros
        left = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]]) if w * ch > ch else np.zeros(w * ch, np.int32)
        diag = np.concatenate([np.zeros(ch, np.int32), up[:-ch]]) if w * ch > ch else np.zeros(w * ch, np.int32)
        p = left + up - diag
        pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - diag)
        paeth = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, diag))
        preds = [zeros, left, up, (left + up) >> 1, paeth]
        res = [((cur - q) & 255) for q in preds]
        if filter_flags is None or y == 0:
            sums = [int(np.where(r < 128, r, 256 - r).sum()) for r in res]
            f = sums.index(min(sums))
        else:
            f = flag_to_id[int(filter_flags[y])]
        ids[y] = f
        rows[y] = res[f].astype(np.uint8)
    return ctype, ids, rows


# ---- CPU run of the GPU deflate core (tests/c/deflate_host.cpp) -------------------------------------------------
_dfl = None


def deflate_host_lib():
    """tests/c/deflate_host.cpp built into a shared object (cached per process)."""
    global _dfl
    if _dfl is None:
        import subprocess
        import tempfile
        so = os.path.join(tempfile.mkdtemp(prefix="dfl_host_"), "libdeflate_host.so")
        subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "c", "deflate_host.cpp"), "-lz", "-lpthread"], check=True)
        lib = C.CDLL(so)
        lib.dfl_host_zlib.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.dfl_host_zlib.restype = C.c_size_t
        lib.dfl_host_set_team.argtypes = [C.c_int]
        _dfl = lib
    return _dfl


def deflate_host(data, max_chain=64, min_len=6, block_bytes=262144, team=0):
    """zlib stream of `data` from the CPU run of the encoder; returns (bytes, stats[stored, fixed, dynamic, tokens]).
    team=0: the one-thread dfl_encode_block (pl_deflate_core.h); team=N: dfl_encode_block_coop (pl_deflate_coop.h, what
    the GPU runs) with a team of N host threads."""
    lib = deflate_host_lib()
    lib.dfl_host_set_team(team)
    src = np.frombuffer(bytes(data), np.uint8) if len(data) else np.zeros(0, np.uint8)
    out = np.zeros(len(data) + len(data) // 4 + 4096, np.uint8)
    stats = np.zeros(4, np.uint32)
    n = lib.dfl_host_zlib(src.ctypes.data if len(data) else None, len(data), out.ctypes.data, out.size, max_chain, min_len,
                          block_bytes, stats.ctypes.data)
    assert n > 0
    return out[:n].tobytes(), stats


def zlib9_filtered(data):
    """zlib the way libpng drives it for the reference's writer: level 9, memLevel 9, Z_FILTERED, 32 KiB window"""
    import zlib
    c = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FILTERED)
    return c.compress(data) + c.flush()


_seg = None


def seg_host_lib():
    """tests/c/seg_host.cpp (the kernel bodies of the segment-parallel row engine, pngloss_amd/csrc/pl_seg_core.h, run as
    loops on the CPU) built into a shared object (cached per process)."""
    global _seg
    if _seg is None:
        import subprocess
        import tempfile
        so = os.path.join(tempfile.mkdtemp(prefix="seg_host_"), "libseg_host.so")
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-o", so, os.path.join(ROOT, "tests", "c", "seg_host.cpp")], check=True)
        lib = C.CDLL(so)
        lib.seg_host_optimize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint, C.c_long, C.c_void_p]
        lib.seg_host_optimize.restype = C.c_int
        _seg = lib
    return _seg


def run_seg_host(img, s=19, b=2, filters=True):
    """The segment-parallel engine's kernel bodies on the CPU: returns rc, out, filters, stats
    (attempts, restarts, retried rows, serial rows, unique symbols, bpp, chain states, status)."""
    out = np.ascontiguousarray(img).copy()
    h, w, _ = out.shape
    f = np.zeros(h, np.uint8)
    st = np.zeros(8, np.uint32)
    rc = seg_host_lib().seg_host_optimize(out.ctypes.data, w, h, f.ctypes.data if filters else None, s, b, st.ctypes.data)
    return rc, out, (f if filters else None), st


_prh = None


def pngread_host_lib():
    """tests/c/pngread_host.cpp (the pixel arithmetic of the device PNG reader, pl_pngread_core.h, on the CPU) as a shared object."""
    global _prh
    if _prh is None:
        import subprocess
        import tempfile
        so = os.path.join(tempfile.mkdtemp(prefix="pngread_host_"), "libpngread_host.so")
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-o", so, os.path.join(ROOT, "tests", "c", "pngread_host.cpp")], check=True)
        lib = C.CDLL(so)
        lib.pngread_host_decode.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_void_p]
        lib.pngread_host_decode.restype = C.c_int
        _prh = lib
    return _prh


def png_read_fixtures():
    """(name, png bytes, expected RGBA8 of the REAL reference reader): tests/golden/png_read_cases.npz + the suite files."""
    g = load_npz("png_read_cases.npz")
    out = [(k[:-4], g[k].tobytes(), g[k[:-4] + "/rgba"]) for k in g.files if k.endswith("/png")]
    s, inp = load_npz("suite_png.npz"), load_npz("suite_inputs.npz")
    out += [("suite_" + k, s[k].tobytes(), inp[k]) for k in s.files]
    return out


def fuzz_case(rng, big=False, strength=None):
    """One random case of the parity campaigns (tests/tools/gpu_fuzz.py on the GPU box by hand, tests/test_gpu_parity.py::test_randomised_parity_campaign in the
    driver's -m gpu run): shapes around the segment / wave sizes, nine kinds of content, six alpha / gray classes, strengths 0..255, bleeds 1..32767, both
    row_filters modes.  Returns (rgba, strength, bleed, want_filters)."""
    w = int(rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 129, 257, int(rng.integers(1, 400))]))
    h = int(rng.integers(1, 30))
    if big:
        w = int(rng.integers(200, 1500)); h = int(rng.integers(20, 120))
    kind = int(rng.integers(0, 9))
    if kind == 0:
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    elif kind == 1:      # smooth gradients + small noise
        x = np.linspace(0, 255, w)[None, :, None]; y = np.linspace(0, 255, h)[:, None, None]
        img = np.clip((x * rng.random(4) + y * rng.random(4)) / 2 + rng.integers(0, 6, (h, w, 4)), 0, 255).astype(np.uint8)
    elif kind == 2:      # saturated: lots of 0 and 255
        img = (rng.integers(0, 2, (h, w, 4)) * 255).astype(np.uint8)
    elif kind == 3:      # near-white with noise (clamping at 255)
        img = (255 - rng.integers(0, 12, (h, w, 4))).astype(np.uint8)
    elif kind == 4:      # near-black
        img = rng.integers(0, 12, (h, w, 4), dtype=np.uint8)
    elif kind == 5:      # few distinct values -> many histogram ties
        img = (rng.integers(0, 3, (h, w, 4)) * 100 + 20).astype(np.uint8)
    elif kind == 6:      # constant
        img = np.full((h, w, 4), int(rng.integers(0, 256)), np.uint8)
    elif kind == 7:      # blocks
        img = np.repeat(np.repeat(rng.integers(0, 256, ((h + 3) // 4, (w + 3) // 4, 4), dtype=np.uint8), 4, 0), 4, 1)[:h, :w]
    else:                # stripes
        img = np.zeros((h, w, 4), np.uint8); img[:, ::2] = rng.integers(0, 256, 4); img[:, 1::2] = rng.integers(0, 256, 4)
    img = np.ascontiguousarray(img)
    cls = int(rng.integers(0, 6))
    if cls == 1: img[..., 3] = 255
    elif cls == 2: img[..., 0] = img[..., 1]; img[..., 2] = img[..., 1]
    elif cls == 3: img[..., 0] = img[..., 1]; img[..., 2] = img[..., 1]; img[..., 3] = 255
    elif cls == 4: img[..., 3] = np.where(rng.random((h, w)) < 0.4, 0, img[..., 3])
    s = int(rng.choice([0, 1, 2, 5, 7, 8, 15, 16, 19, 20, 23, 24, 31, 32, 40, 47, 48, 63, 64, 85, 100, 127, 200, 255, int(rng.integers(0, 256))]))
    b = int(rng.choice([1, 2, 3, 4, 8, 16, 100, 1000, 32767, int(rng.integers(1, 32768))]))
    if strength is not None: s = int(strength)
    return img, s, b, bool(rng.integers(0, 3))
