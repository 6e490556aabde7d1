"""The segment-parallel row engine (pngloss_amd/csrc/pl_seg_core.h: enumerate / chain / replay / validate / control) run on the
CPU by tests/c/seg_host.cpp -- the same kernel bodies hipcc compiles, as loops over (workgroup, thread) -- against the oracle.
Bit-exact bytes and filter IDs on every byte-per-pixel class, strengths with few and many chain states, bleed dividers,
rows that need the strength retry, NULL row_filters (every row adaptive), ragged widths around the segment and group sizes."""
import numpy as np
import pytest

import pngloss_amd as P
from tests import util as U

CASES = [(64, 48, m, 19, 2) for m in range(6)] + [
    (200, 40, 0, 19, 2), (333, 37, 1, 19, 2), (100, 30, 0, 20, 8), (97, 33, 2, 7, 3), (130, 20, 5, 19, 4),
    (1, 1, 1, 19, 2), (2, 3, 1, 19, 2), (5, 1, 1, 19, 2), (1, 7, 1, 19, 2), (33, 5, 3, 19, 2), (64, 6, 4, 0, 2),
    (31, 9, 0, 19, 2), (32, 9, 0, 19, 2), (65, 9, 5, 19, 2), (511, 6, 0, 19, 2), (513, 6, 1, 19, 2), (96, 20, 0, 3, 1),
    (80, 12, 1, 19, 32767),
    # state sets enumerated in several chunks of lanes (259 .. 955 states)
    (260, 24, 0, 20, 2), (200, 24, 1, 20, 1), (200, 20, 2, 40, 2), (200, 20, 0, 85, 8), (150, 20, 4, 26, 2), (150, 16, 3, 40, 2),
    # state sets beyond the lanes (10^4 .. 10^5 chain states): the SEEDED enumeration (run-in from seeds, entry states looked up by value)
    (200, 20, 0, 85, 1), (200, 20, 1, 85, 2), (200, 16, 5, 40, 1), (150, 12, 3, 255, 1), (130, 16, 2, 160, 1), (97, 12, 4, 200, 3), (300, 10, 0, 128, 2),
    (33, 6, 1, 85, 1), (1, 3, 1, 85, 1), (64, 6, 5, 255, 2),
    # rows beyond 8192 pixels: the chain kernel takes them in passes
    (8300, 2, 0, 19, 2), (8300, 2, 1, 85, 1),
]


@pytest.mark.parametrize("w,h,mode,s,b", CASES)
def test_seg_engine_bodies_match_oracle(w, h, mode, s, b):
    img = P.synth_rgba(w, h, mode, 0)
    rc, out, f, st = U.run_seg_host(img, s, b)
    assert rc == 0, "more chain states than lanes? stats %s" % st
    want, wf = U.run_port(img, s, b)
    assert np.array_equal(out, want) and np.array_equal(f, wf)


def test_seg_engine_speculation_is_right_almost_always():
    """The validation pass makes every result exact whatever the speculation did -- a bug in the tables, the maps or the chain shows
    only as extra attempts.  So the attempt count is pinned: a 1024-wide frame needs one attempt per row plus a handful of epochs
    (measured: 138 attempts for 128 rows, 5 epochs -- an epoch costs two attempts since the validation runs one launch behind: the attempt
    under way when it fails is void; candidate none is ruled out by its cost bound in most rows)."""
    img = P.synth_rgba(1024, 128, 0, 0)
    rc, out, f, st = U.run_seg_host(img, 19, 2)
    want, wf = U.run_port(img, 19, 2)
    assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf)
    attempts, restarts, serial = int(st[0]), int(st[1]), int(st[3])
    assert attempts <= 128 + 24 and restarts <= 12 and serial == 0, (attempts, restarts, serial)


@pytest.mark.parametrize("nt", [512, 1024])
@pytest.mark.parametrize("w,h,mode,s,b", [(333, 37, 1, 19, 2), (520, 24, 0, 19, 2), (300, 24, 1, 20, 1), (200, 20, 4, 26, 2), (97, 33, 2, 7, 3), (64, 6, 4, 0, 2)])
def test_seg_engine_both_sizes_of_the_enumeration_workgroups(monkeypatch, nt, w, h, mode, s, b):
    """the enumeration's workgroups have 512 threads (a channel pair) for rows up to 2560 pixels and 1024 (all four channels) beyond;
    the harness follows the same rule, SEG_HOST_ENUM_NT pins one"""
    monkeypatch.setenv("SEG_HOST_ENUM_NT", str(nt))
    img = P.synth_rgba(w, h, mode, 0)
    rc, out, f, st = U.run_seg_host(img, s, b)
    want, wf = U.run_port(img, s, b)
    assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf)


@pytest.mark.parametrize("s,b,most", [(40, 2, 124), (85, 8, 118), (20, 1, 130)])
def test_seg_engine_speculation_with_state_sets_enumerated_in_chunks(s, b, most):
    """the same pin for state sets of 650 .. 955 chain states (measured: 112 / 106 / 118 attempts for 96 rows, 10 / 3 / 11 epochs of two attempts each): the chunked
    enumeration, the dense transition tables and the wide table stride are right when the attempts stay near one per row"""
    img = P.synth_rgba(1024, 96, 0, 0)
    rc, out, f, st = U.run_seg_host(img, s, b)
    want, wf = U.run_port(img, s, b)
    assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf)
    assert int(st[0]) <= most and int(st[3]) == 0, st


@pytest.mark.parametrize("flags", [2, 4, 6])
@pytest.mark.parametrize("w,h,mode,s,b", [(333, 37, 3, 19, 2), (520, 24, 0, 19, 2), (300, 24, 1, 20, 1), (200, 20, 4, 26, 2), (300, 12, 0, 85, 1), (260, 10, 5, 255, 1)])
def test_seg_engine_chain_fallback_paths(monkeypatch, flags, w, h, mode, s, b):
    """the chain kernel's rare paths, forced through its test hooks: 2 = every second segment through the repair (a segment whose entry
    state the enumeration did not cover is walked step by step, the passes go on behind it), 4 = a wider table stride than needed"""
    monkeypatch.setenv("SEG_HOST_FLAGS", str(flags))
    img = P.synth_rgba(w, h, mode, 0)
    rc, out, f, st = U.run_seg_host(img, s, b)
    want, wf = U.run_port(img, s, b)
    assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf)


def test_seg_engine_all_rows_adaptive():
    img = P.synth_rgba(150, 24, 0, 3)
    rc, out, f, st = U.run_seg_host(img, 19, 2, filters=False)
    want, _ = U.run_port(img, 19, 2, filters=False)
    assert rc == 0 and np.array_equal(out, want)


def test_seg_engine_takes_every_strength_and_bleed():
    """no (strength, bleed) pair is declined any more: the sets that do not fit the lanes are enumerated from seeds"""
    img = P.synth_rgba(40, 8, 0, 0)
    for s, b in [(85, 2), (44, 2), (28, 1), (127, 1), (128, 1), (255, 1), (255, 32767), (119, 8)]:
        rc, out, f, st = U.run_seg_host(img, s, b)
        want, wf = U.run_port(img, s, b)
        assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf), (s, b)


@pytest.mark.parametrize("s,b,most", [(85, 1, 74), (85, 2, 64), (40, 1, 108)])
def test_seg_engine_seeded_speculation_is_right_almost_always(s, b, most):
    """the seeded enumeration's entry sets hold the reference's own state nearly always (measured: 66 / 58 / 98 attempts for 48 rows with 9 / 3 / 30 epochs of two attempts each,
    at most a handful of segments walked step by step by the chain kernel): wrong seeds, a wrong hash or a wrong lookup would show
    as attempts or repairs, never as wrong bytes"""
    img = P.synth_rgba(1024, 48, 0, 0)
    rc, out, f, st = U.run_seg_host(img, s, b)
    want, wf = U.run_port(img, s, b)
    assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf)
    assert int(st[0]) <= most and int(st[3]) == 0 and int(st[6]) == 0, st


@pytest.mark.parametrize("w,h,mode,s,b", [(333, 20, 1, 19, 2), (520, 12, 0, 19, 2), (200, 12, 5, 7, 3), (97, 9, 3, 0, 2), (150, 10, 2, 20, 1)])
def test_seg_engine_seeded_enumeration_at_small_strengths(monkeypatch, w, h, mode, s, b):
    """SEG_HOST_SEEDED=1 makes the harness use the seeded enumeration where the exhaustive one would do: same bytes"""
    monkeypatch.setenv("SEG_HOST_SEEDED", "1")
    img = P.synth_rgba(w, h, mode, 0)
    rc, out, f, st = U.run_seg_host(img, s, b)
    want, wf = U.run_port(img, s, b)
    assert rc == 0 and int(st[6]) == 0 and np.array_equal(out, want) and np.array_equal(f, wf)


def test_seg_engine_seeded_random_images():
    """random images of every class incl. transparency and few-valued ones (ties, saturation), the strengths/bleeds the lanes cover"""
    ran = 0
    for img, s, b, want_filters in U.seeded_cases(seed=11, n=36):
        rc, out, f, st = U.run_seg_host(img, s, b, filters=want_filters)
        if rc == 64:
            continue
        want, wf = U.run_port(img, s, b, filters=want_filters)
        assert rc == 0 and np.array_equal(out, want), (img.shape, s, b, st)
        if want_filters:
            assert np.array_equal(f, wf)
        ran += 1
    assert ran >= 12


def test_seg_engine_bodies_clean_under_asan_and_ubsan(tmp_path):
    """the same kernel bodies built with -fsanitize=address,undefined (every device array of the harness is its own heap block, so an
    index past an array is a report): a few shapes incl. the widest row the engine takes and state sets enumerated in chunks, run in
    a child process with the sanitizer runtime preloaded; no report, results equal to the oracle's"""
    import os
    import subprocess
    import sys
    so = tmp_path / "libseg_host_san.so"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-fno-omit-frame-pointer", "-w", "-o", str(so), os.path.join(U.ROOT, "tests", "c", "seg_host.cpp")], check=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    code = (
        "import ctypes as C, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import pngloss_amd as P\n"
        "from tests import util as U\n"
        "lib = C.CDLL(%r)\n"
        "lib.seg_host_optimize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint, C.c_long, C.c_void_p]\n"
        "for (w, h, m, s, b) in [(700, 10, 0, 19, 2), (300, 8, 2, 40, 2), (8192, 2, 0, 19, 2), (333, 7, 3, 19, 2), (33, 5, 5, 7, 3), (3300, 3, 1, 20, 1), (700, 6, 0, 85, 1), (333, 6, 5, 255, 1), (8192, 2, 1, 85, 2)]:\n"
        "    img = P.synth_rgba(w, h, m, 0); out = img.copy(); f = np.zeros(h, np.uint8); st = np.zeros(8, np.uint32)\n"
        "    rc = lib.seg_host_optimize(out.ctypes.data, w, h, f.ctypes.data, s, b, st.ctypes.data)\n"
        "    want, wf = U.run_port(img, s, b)\n"
        "    assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf), (w, h, m, s, b)\n"
        "import os\n"
        "for unit in ('1', '0'):\n"      # round 6: the enumeration FROM SEEDS, in units and segment by segment (its run-in records and staged maps live behind the first phase's scratch)
        "    os.environ['SEG_HOST_UNIT'] = unit; os.environ['SEG_HOST_SEEDS'] = '1'\n"
        "    for (w, h, m, s, b) in [(700, 10, 0, 19, 2), (1920, 6, 0, 19, 2), (333, 7, 3, 19, 2), (33, 5, 5, 7, 3), (513, 8, 5, 19, 2), (2100, 3, 1, 12, 1), (8192, 2, 0, 19, 2)]:\n"
        "        img = P.synth_rgba(w, h, m, 0); out = img.copy(); f = np.zeros(h, np.uint8); st = np.zeros(8, np.uint32)\n"
        "        rc = lib.seg_host_optimize(out.ctypes.data, w, h, f.ctypes.data, s, b, st.ctypes.data)\n"
        "        want, wf = U.run_port(img, s, b)\n"
        "        assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf), ('seeds', unit, w, h, m, s, b)\n"
        "print('sanitized ok')\n") % (U.ROOT, str(so))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "sanitized ok" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]



@pytest.mark.parametrize("w,h,mode,s,b", [(333, 37, 1, 19, 2), (600, 12, 5, 19, 2), (520, 16, 3, 19, 2), (200, 20, 0, 85, 1), (97, 33, 2, 7, 3), (1, 7, 1, 19, 2), (150, 12, 3, 255, 1)])
def test_seg_engine_validation_and_control_in_either_order(monkeypatch, w, h, mode, s, b):
    """the validation of an attempt runs in the SAME launch as the control workgroups that decide on it (optimistically), so neither may read
    what the other writes: the harness runs the two halves of that launch one after the other -- here validation first, everywhere else
    control first; a hazard (the committed row, the shifted error rows, the next control block seen by the validation, or its verdict seen
    by the decision) would change bytes or attempts in one of the orders"""
    img = P.synth_rgba(w, h, mode, 0)
    rc0, out0, f0, st0 = U.run_seg_host(img, s, b)
    monkeypatch.setenv("SEG_HOST_VAL_FIRST", "1")
    rc, out, f, st = U.run_seg_host(img, s, b)
    want, wf = U.run_port(img, s, b)
    assert rc == 0 and rc0 == 0 and np.array_equal(out, want) and np.array_equal(f, wf) and np.array_equal(out0, want)
    assert list(st) == list(st0)


@pytest.mark.parametrize("key,s,b,filters", [("r4_many_attempts_s200_b32767_null", 200, 32767, False), ("r4_many_attempts_s255_b3_ids", 255, 3, True)])
def test_seg_engine_tiny_images_that_need_very_many_attempts(key, s, b, filters):
    """Two cases of the round-4 parity campaign on the GPU box (tests/tools/gpu_fuzz.py, seed 42): 63 x 2 and 17 x 2 pixels at strengths 200 / 255,
    where nearly every row attempt fails validation (a histogram of a few counts: every bump flips a tie), every strength down to the one that
    passes libpng's heuristic is tried, and an epoch costs two attempts -- 1813 and 1267 attempts.  The launcher's bound on the attempts (a
    runaway stop, not a budget) had not counted the strength retries: the library gave up with PNGLOSS_HIP_ERROR.  Inputs: tests/golden/fuzz_regressions.npz."""
    img = U.load_npz("fuzz_regressions.npz")[key]
    rc, out, f, st = U.run_seg_host(img, s, b, filters)
    want, wf = U.run_port(img, s, b, filters)
    assert rc == 0 and np.array_equal(out, want) and (not filters or np.array_equal(f, wf)), st


UNIT_CASES = [(64, 48, m, 19, 2) for m in range(6)] + [
    (200, 40, 0, 19, 2), (333, 37, 1, 19, 2), (130, 20, 5, 19, 4), (1, 1, 1, 19, 2), (5, 1, 1, 19, 2), (1, 7, 1, 19, 2), (33, 5, 3, 19, 2), (97, 33, 2, 7, 3),
    (127, 9, 0, 19, 2), (128, 9, 0, 19, 2), (129, 9, 4, 19, 2), (511, 6, 0, 19, 2), (513, 6, 1, 19, 2), (640, 8, 3, 19, 2), (96, 20, 0, 3, 1), (80, 12, 1, 19, 32767),
    (1600, 6, 0, 19, 2), (8300, 2, 0, 19, 2),
    (95, 7, 0, 19, 2), (191, 5, 1, 19, 2), (192, 5, 0, 19, 2), (193, 5, 5, 19, 2), (289, 4, 2, 19, 2), (384, 4, 3, 19, 2),     # around SEG_UNIT * 32 pixels and the twelve pairs of a workgroup
]


@pytest.mark.parametrize("w,h,mode,s,b", UNIT_CASES)
def test_seg_engine_enumeration_in_units_matches_oracle(monkeypatch, w, h, mode, s, b):
    """What the launcher asks for when a BATCH keeps the GPU busy (SegParams::unit = SEG_UNIT; pl_host.hip): runs of four segments enumerated as one unit
    -- from every state only at the unit's first pixel, the distinct states through all its segments, twelve (unit, channel) pairs per workgroup, none / up
    through the same body with their own small state set -- and the chain kernel composing units.  Widths around the unit (128 pixels) and the
    workgroup's twelve pairs, every byte-per-pixel class, rows that need epochs (their first unit is walked) and the strength retry."""
    monkeypatch.setenv("SEG_HOST_UNIT", "1")
    img = P.synth_rgba(w, h, mode, 0)
    rc, out, f, st = U.run_seg_host(img, s, b)
    want, wf = U.run_port(img, s, b)
    assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf)


@pytest.mark.parametrize("w,h,mode,s,b", UNIT_CASES + [(1920, 24, 0, 19, 2), (700, 30, 1, 19, 2), (513, 20, 5, 19, 2), (900, 16, 0, 12, 1), (1600, 10, 0, 7, 3), (640, 12, 0, 31, 8)])
def test_seg_engine_units_from_seeds_match_oracle(monkeypatch, w, h, mode, s, b):
    """Round 6: the units of a batch start FROM SEEDS -- one per left byte within reach of the data, a run-in of eight pixels in front of the unit -- instead of from every
    state at the unit's first pixel (seg_enum_unit_body<.., SEEDS = true>; SEG_HOST_SEEDS=1 is what the launcher offers for every (strength, bleed) pair with a seed set).
    Every class, widths around a unit and around a workgroup's sixteen pairs, rows that need epochs (started exhaustively), the strength retry, transparent pixels in the
    run-in, pairs whose seeds fall outside 0..255."""
    monkeypatch.setenv("SEG_HOST_UNIT", "1")
    monkeypatch.setenv("SEG_HOST_SEEDS", "1")
    img = P.synth_rgba(w, h, mode, 0)
    rc, out, f, st = U.run_seg_host(img, s, b)
    want, wf = U.run_port(img, s, b)
    assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf)


@pytest.mark.parametrize("w,h,mode,s,b", [(1024, 24, 0, 19, 2), (700, 30, 1, 19, 2), (513, 20, 5, 19, 2), (385, 20, 2, 19, 2), (289, 20, 3, 19, 2), (193, 20, 4, 19, 2), (95, 7, 0, 19, 2),
                                          (33, 9, 1, 19, 2), (1600, 10, 0, 7, 3), (900, 16, 0, 12, 1), (640, 12, 0, 31, 8), (2100, 5, 0, 19, 2)])
def test_seg_engine_segments_from_seeds_match_oracle(monkeypatch, w, h, mode, s, b):
    """Round 6, small and mid-size batches (pl_seg.hip:seg_k_enum_unit<1>): the unit enumeration's bodies SEGMENT BY SEGMENT -- (segment, channel) pairs, sixteen a
    workgroup, each started from seeds eight pixels in front of it -- with the per-segment chain, replay and control kernels of one image."""
    monkeypatch.setenv("SEG_HOST_UNIT", "0")
    monkeypatch.setenv("SEG_HOST_SEEDS", "1")
    img = P.synth_rgba(w, h, mode, 0)
    rc, out, f, st = U.run_seg_host(img, s, b)
    want, wf = U.run_port(img, s, b)
    assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf)


def test_seed_sets_of_the_unit_enumeration():
    """The seeds a unit starts from (pl_seg_core.h:seg_build_exhaustive, round 6): one member of the exhaustive state list per left byte within reach of the data (delta = -dmax ..
    dmax, each once), with the carried terms of the diff that explains it when as little as possible was carried into the boundary pixel -- for |delta| <= s exactly the split of
    -delta (optimize_state.c:445-467: rem, threes).  s = 19 b = 2: 47 of 253 states; no seed set where the reach does not fit 64 lanes, the set has more than 255 states, or it is seeded anyway."""
    import ctypes as C
    lib = U.seg_host_lib()
    lib.seg_host_seeds.argtypes = [C.c_uint, C.c_long, C.c_void_p, C.c_void_p]
    lib.seg_host_seeds.restype = C.c_int

    def seeds(s, b):
        out = np.zeros(4, np.int32); st = np.zeros(3 * 64, np.int32)
        assert lib.seg_host_seeds(s, b, out.ctypes.data, st.ctypes.data) == 0
        return [int(v) for v in out], st.reshape(64, 3)[: out[0]]

    (n, kin, ns, dmax), st = seeds(19, 2)
    assert (n, kin, ns, dmax) == (47, 8, 253, 23)
    assert sorted(int(d) for d in st[:, 0]) == list(range(-23, 24))
    for delta, cn, th in st:
        if abs(delta) <= 19:
            d = int(-delta) // 2 if -delta >= 0 else -(int(delta) // 2)          # truncating division by the bleed divider
            t = int(d / 16); d -= 4 * t; h = int(d / 8); d -= 2 * h; f = int(d * 2 / 9); d -= 2 * f; v = int(d / 2); d -= v
            assert (cn, th) == (d, h), (delta, cn, th)
    for (s, b), want in {(7, 3): True, (12, 1): True, (31, 8): True, (20, 2): False, (40, 2): False, (40, 8): False, (85, 1): False, (20, 1): True}.items():
        (n, kin, ns, dmax), _ = seeds(s, b)
        assert (n > 0) == (want and 2 * dmax + 1 <= 64 and 0 < ns <= 255), (s, b, n, ns, dmax)


def test_seg_engine_units_from_seeds_cost_few_attempts_and_a_stuck_row_one_break(monkeypatch):
    """What a seed set that misses a state costs is attempts, and the count is pinned: on photographic rows the seeds find every entry state (as many attempts as the
    start from every state, +-2); the 60-row frame of the generator has ONE row whose alpha channel -- a sawtooth of period 32 -- keeps candidate sub in a cycle no seed
    reaches at ten unit boundaries in a row: the row is broken off ONCE and its epoch starts from every state (seg_unit_from_seeds), so the frame takes 74 attempts
    against 70 -- not 92, which is what ten breaks cost before epochs started exhaustively."""
    monkeypatch.setenv("SEG_HOST_UNIT", "1")
    res = {}
    for (w, h) in [(1024, 96), (1920, 60)]:
        img = P.synth_rgba(w, h, 0, 0)
        want, wf = U.run_port(img, 19, 2)
        for seeds in ("0", "1"):
            monkeypatch.setenv("SEG_HOST_SEEDS", seeds)
            rc, out, f, st = U.run_seg_host(img, 19, 2)
            assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf)
            res[(w, seeds)] = int(st[0])
    assert abs(res[(1024, "1")] - res[(1024, "0")]) <= 2, res
    assert res[(1920, "1")] <= res[(1920, "0")] + 6, res


def test_seg_engine_units_from_seeds_fall_back_on_flat_content(monkeypatch):
    """Flat, few-coloured content is full of fixed points the seeds do not reach (oracle/seed_study.c on the suite's dice and tux: one unit boundary in a hundred): an image
    whose rows keep breaking goes back to the start from every state for good (seg_unit_from_seeds: more than eight breaks and one row in sixteen) -- exact either way, and
    the attempts stay within a quarter of the exhaustive start's."""
    monkeypatch.setenv("SEG_HOST_UNIT", "1")
    g = U.load_npz("suite_small.npz")
    img = np.ascontiguousarray(g["tux/in"])
    want, wf = g["tux/out"], g["tux/filters"]
    res = {}
    for seeds in ("0", "1"):
        monkeypatch.setenv("SEG_HOST_SEEDS", seeds)
        rc, out, f, st = U.run_seg_host(img, 19, 2)
        assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf)
        res[seeds] = int(st[0])
    assert res["1"] <= res["0"] * 5 // 4 + 8, res


@pytest.mark.parametrize("filters", [True, False])
def test_seg_engine_units_cost_no_attempts(monkeypatch, filters):
    """the attempt count is what a wrong map, id or entry state would show in (the validation makes the bytes right whatever happens): enumeration in units
    takes exactly the attempts the per-segment enumeration takes"""
    img = P.synth_rgba(1024, 96, 0, 0)
    want, wf = U.run_port(img, 19, 2, filters=filters)
    res = {}
    for unit in ("0", "1"):
        monkeypatch.setenv("SEG_HOST_UNIT", unit)
        rc, out, f, st = U.run_seg_host(img, 19, 2, filters=filters)
        assert rc == 0 and np.array_equal(out, want)
        if filters:
            assert np.array_equal(f, wf)
        res[unit] = (int(st[0]), int(st[1]), int(st[3]))
    assert res["1"][0] <= res["0"][0] + 2 and res["1"][2] == 0, res


def test_seg_engine_units_with_a_state_set_of_several_chunks(monkeypatch):
    """(the launcher does not pick units for such sets -- the distinct states of twelve pairs outgrow a workgroup's lanes and the surplus costs epochs --,
    but the body must stay exact there: SEG_HOST_UNIT=2 forces it)"""
    monkeypatch.setenv("SEG_HOST_UNIT", "2")
    for (w, h, mode, s, b) in [(260, 12, 0, 20, 2), (200, 10, 2, 40, 2)]:
        img = P.synth_rgba(w, h, mode, 0)
        rc, out, f, st = U.run_seg_host(img, s, b)
        want, wf = U.run_port(img, s, b)
        assert rc == 0 and np.array_equal(out, want) and np.array_equal(f, wf)
