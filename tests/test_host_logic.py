"""CPU tests of the host logic: the C ABI surface, the exact-division trick the kernels rely on, the synthetic
generator and the sharding helpers.  No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pngloss_amd as P
from pngloss_amd import lib as L
from pngloss_amd import shard as S
from tests import util as U


def test_library_loads_and_exports_every_declared_symbol():
    header = open(os.path.join(U.ROOT, "include", "pngloss_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", header))
    declared -= {"defined"}
    assert declared == set(L.ABI_SYMBOLS), declared ^ set(L.ABI_SYMBOLS)
    lib = P.hip_lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.pngloss_hip_version()


def test_no_silent_cpu_fallback_without_a_gpu():
    lib = P.hip_lib()
    n = lib.pngloss_hip_device_count()
    if n > 0:
        pytest.skip("a GPU is present; the loud-failure path is exercised on GPU-less boxes")
    assert n <= 0
    assert not lib.pngloss_hip_create(-1)
    img = P.synth_rgba(8, 4, 0, 0)
    before = img.copy()
    with pytest.raises(RuntimeError):
        P.optimize_with_rows(img, 19, 2)
    assert np.array_equal(img, before)
    with pytest.raises(RuntimeError):
        P.HipContext()


def test_product_code_never_touches_the_oracle():
    """only tests/, bench.py (cpu_baseline) and __graft_entry__.smoke() may use oracle/."""
    for dirpath, _, files in os.walk(os.path.join(U.ROOT, "pngloss_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".c", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "libpngloss_port" not in text and "libpngloss_ref" not in text and "pngloss_port.h" not in text, fn
    text = open(os.path.join(U.ROOT, "include", "pngloss_hip.h")).read()
    assert "port_" not in text


def test_no_environment_variable_of_the_shipped_library_changes_results():
    """Round 6 (the round-5 review's item): the one hook that changed results -- "candidate f wins every row", PNGLOSS_HIP_FORCE_FILTER, a debugging aid of rounds 1-3 -- is a
    BUILD flag now (-DPL_DEBUG_FORCE_FILTER=f): the shipped library does not know the variable, its version string does not announce a debugging build, and the
    environment hooks that are left (timing and test pins) are read in one place, when a context is created -- PNGLOSS_HIP_ENGINE, the tests' pin of the row engine,
    is the one read per call."""
    so = open(os.path.join(U.ROOT, "pngloss_amd", "csrc", "libpngloss_hip.so"), "rb").read()
    assert b"PNGLOSS_HIP_FORCE_FILTER" not in so
    assert b"DEBUGGING BUILD" not in P.hip_lib().pngloss_hip_version()
    host = open(os.path.join(U.ROOT, "pngloss_amd", "csrc", "pl_host.hip")).read()
    body = host[host.index("static PlHooks from_env()"):]
    body = body[body.index("return h;"):]              # (everything behind the hooks' reader)
    import re
    left = set(re.findall(r'getenv\("([A-Z_]+)"\)', body))
    assert left == {"PNGLOSS_HIP_ENGINE", "PNGLOSS_DEVICES"}, left


def _recip_up(d, ulps=1):
    r = np.float32(1.0) / np.float32(d)
    for _ in range(ulps):
        r = np.nextafter(r, np.float32(2.0))
    return np.float32(r)


@pytest.mark.parametrize("ulps", [1, 2])
def test_float_reciprocal_division(ulps):
    """pl_device.h:pl_truncdiv_f -- trunc(float(n) * recip_up(d)) == n // d for every n the kernels can produce
    (|n| < 2^17) and every divisor they use: q = strength+1 in 1..256, the bleed divider 1..32767, and 9."""
    n = np.arange(0, 1 << 17, dtype=np.int64)
    nf = n.astype(np.float32)
    divisors = list(range(1, 257)) + [257, 1000, 4095, 4096, 4097, 9999, 16384, 32766, 32767]
    for d in divisors:
        got = np.trunc(nf * _recip_up(d, ulps)).astype(np.int64)
        assert np.array_equal(got, n // d), d
    # the 2d/9 step of the Sierra split uses 2*recip_up(9) on d directly
    r29 = np.float32(2.0) * _recip_up(9, ulps)
    assert np.array_equal(np.trunc(nf * r29).astype(np.int64), (2 * n) // 9)
    # negative operands: C truncation toward zero is symmetric
    assert np.array_equal(np.trunc(-nf * _recip_up(7, ulps)).astype(np.int64), -(n // 7))


def test_rank_preserves_order_and_equality():
    rng = np.random.default_rng(2)
    o = rng.integers(0, 40, 256)
    rank = np.array([(o < v).sum() for v in o])
    assert rank.max() <= 255
    for a in range(0, 256, 7):
        for b in range(256):
            assert (o[a] < o[b]) == (rank[a] < rank[b]) and (o[a] == o[b]) == (rank[a] == rank[b])


def test_synthetic_generator_matches_survey_digests():
    for e in U.load_digests()["synthetic"]:
        if e["width"] * e["height"] <= 1920 * 1080:
            img = P.synth_rgba(e["width"], e["height"], e["mode"], e["frame"])
            assert "%016x" % P.fnv1a64(img, P.SURVEY_FNV_BASIS) == e["in"]
    a = P.synth_rgba(33, 17, 0, 5)
    assert np.array_equal(a, P.synth_rgba(33, 17, 0, 5))
    assert (P.synth_rgba(16, 16, 2)[..., 3] == 255).all()
    g = P.synth_rgba(16, 16, 4)
    assert (g[..., 0] == g[..., 1]).all() and (g[..., 2] == g[..., 1]).all() and (g[..., 3] == 255).all()
    assert (P.synth_rgba(32, 32, 5)[..., 3] == 0).any()
    # standard FNV-1a-64 known answers
    assert P.fnv1a64(np.frombuffer(b"a", np.uint8)) == 0xAF63DC4C8601EC8C
    assert P.fnv1a64(np.frombuffer(b"", np.uint8)) == 0xCBF29CE484222325


def test_lpt_partition_is_balanced_and_complete():
    rng = np.random.default_rng(0)
    costs = [int(v) for v in rng.integers(1, 1000, 37)]
    for world in (1, 2, 3, 8):
        shards = S.lpt_partition(costs, world)
        assert sorted(i for s in shards for i in s) == list(range(len(costs)))
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(costs)
    assert S.contiguous_partition(256, 8)[3] == list(range(96, 128))
    assert S.contiguous_partition(5, 2) == [[0, 1, 2], [3, 4]]
    assert S.lpt_partition([5, 5, 5, 5], 2) == [[0, 2], [1, 3]]


def test_c_host_split_equals_the_python_split():
    """pngloss_hip_multi_split (the C host's deal of a batch over the GPUs of a node, replaces the sequential file loop of
    /root/reference/src/pngloss.c:173-208) is the same deterministic LPT split as pngloss_amd.shard.lpt_partition."""
    import pngloss_amd as P
    from pngloss_amd import shard as S
    rng = np.random.default_rng(3)
    for trial in range(20):
        n = int(rng.integers(0, 40))
        shapes = [(int(rng.integers(1, 2000)), int(rng.integers(1, 2000))) for _ in range(n)]
        if trial % 3 == 0 and n:
            shapes = [shapes[0]] * n                    # equal sizes: ties broken by index and by part
        for parts in (1, 2, 3, 8):
            own = P.multi_split(shapes, parts)
            want = [None] * n
            for r, items in enumerate(S.lpt_partition([w * h for w, h in shapes], parts)):
                for i in items:
                    want[i] = r
            assert own == want, (trial, parts)
