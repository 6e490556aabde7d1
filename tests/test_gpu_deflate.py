"""GPU deflate (pngloss_hip_optimize_batch_host_zlib) through the C ABI: the zlib stream of every image must inflate
to exactly the scanlines the _emit call returns (i.e. to what libpng would have been handed by the reference,
rwpng.c:477-637), must be byte-identical to the CPU run of the same encoder core (tests/c/deflate_host.cpp), and must
not be larger than zlib level 9 / Z_FILTERED on image-sized inputs."""
import zlib

import numpy as np
import pytest

import pngloss_amd as P
from tests import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = P.HipContext(0)
    yield c
    c.close()


def stream_of(emitted):
    ctype, ids, rows = emitted
    return np.concatenate([ids[:, None], rows], axis=1).tobytes() if rows.size else b""


def run_both(ctx, arrays, s=19, b=2, want_filters=True):
    outs, filts, emitted = ctx.run_host_emit(arrays, s, b, want_filters)
    outs2, filts2, streams = ctx.run_host_zlib(arrays, s, b, want_filters)
    for a, c in zip(outs, outs2):
        assert np.array_equal(a, c)
    return emitted, streams


def test_streams_inflate_to_the_emitted_scanlines_all_modes(ctx):
    arrays = [P.synth_rgba(96, 64, m, f) for m in range(6) for f in range(2)]
    emitted, streams = run_both(ctx, arrays)
    for e, (ctype, z, blocks) in zip(emitted, streams):
        assert ctype == e[0]
        assert zlib.decompress(z) == stream_of(e)
        assert sum(blocks) == 1


def test_streams_are_byte_identical_to_the_cpu_run_of_the_core(ctx):
    arrays = [P.synth_rgba(w, h, m, 3) for (w, h, m) in [(200, 150, 0), (333, 211, 1), (640, 480, 2), (257, 129, 3), (512, 300, 4), (301, 77, 5)]]
    emitted, streams = run_both(ctx, arrays)
    for e, (ctype, z, blocks) in zip(emitted, streams):
        want, stats = U.deflate_host(stream_of(e))
        assert z == want
        assert tuple(int(x) for x in stats[:3]) == blocks


def test_edge_shapes_and_empty_images(ctx):
    shapes = [(1, 1), (2, 3), (5, 1), (1, 7), (700, 1), (1, 700), (3, 3)]
    arrays = [P.synth_rgba(w, h, 0, 2) for (w, h) in shapes] + [np.zeros((0, 5, 4), np.uint8), np.zeros((4, 0, 4), np.uint8)]
    emitted, streams = run_both(ctx, arrays)
    for i, (e, (ctype, z, blocks)) in enumerate(zip(emitted, streams)):
        if arrays[i].size == 0:
            assert z == b""
            continue
        assert zlib.decompress(z) == stream_of(e)
        assert z == U.deflate_host(stream_of(e))[0]


def test_multi_block_images_adaptive_mode_and_strengths(ctx):
    # 1080p RGBA: 8.3 MB of scanlines = 32 deflate blocks; also all-rows-adaptive mode and other strengths
    big = P.synth_rgba(1920, 1080, 0, 0)
    emitted, streams = run_both(ctx, [big])
    data = stream_of(emitted[0])
    ctype, z, blocks = streams[0]
    assert zlib.decompress(z) == data and sum(blocks) == -(-len(data) // 262144)
    assert len(z) <= 0.95 * len(U.zlib9_filtered(data))            # measured: 6.1 % smaller than zlib level 9
    small = [P.synth_rgba(320, 200, 0, 1)]
    for s, b, wf in [(0, 2, True), (40, 2, True), (255, 1, True), (19, 2, False)]:
        emitted, streams = run_both(ctx, small, s, b, wf)
        assert zlib.decompress(streams[0][1]) == stream_of(emitted[0])


def test_incompressible_and_constant_content(ctx):
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (300, 400, 4), dtype=np.uint8)
    flat = np.full((300, 400, 4), 200, np.uint8)
    emitted, streams = run_both(ctx, [noise, flat], 0, 2)           # strength 0: pixels unchanged
    for e, (ctype, z, blocks) in zip(emitted, streams):
        assert zlib.decompress(z) == stream_of(e)
    assert streams[0][2][0] >= 1                                    # noise: stored blocks
    assert len(streams[0][1]) <= len(stream_of(emitted[0])) + 64
    assert len(streams[1][1]) < 2000


def test_mixed_batch_matches_single_image_calls(ctx):
    arrays = [P.synth_rgba(w, h, m, 7) for (w, h, m) in [(64, 64, 0), (1000, 600, 0), (17, 900, 2), (800, 31, 4), (256, 256, 1)]]
    _, batch = run_both(ctx, arrays)
    for a, got in zip(arrays, batch):
        _, single = run_both(ctx, [a])
        assert single[0] == got


def test_batches_split_into_several_groups(tmp_path):
    """A call handles at most 1 GiB of scanlines at a time and walks through larger batches in groups; a small group
    size (test hook PNGLOSS_HIP_DEFLATE_GROUP_BYTES) makes that path run with a handful of images."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, zlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import pngloss_amd as P\n"
        "from tests import util as U\n"
        "ctx = P.HipContext(0)\n"
        "arrays = [P.synth_rgba(w, h, m, 9) for (w, h, m) in [(300, 200, 0), (64, 64, 1), (500, 400, 2), (31, 17, 3), (400, 300, 5), (200, 100, 0), (128, 128, 4)]]\n"
        "outs, filts, emitted = ctx.run_host_emit(arrays)\n"
        "outs2, filts2, streams = ctx.run_host_zlib(arrays)\n"
        "for e, (ctype, z, blocks) in zip(emitted, streams):\n"
        "    data = np.concatenate([e[1][:, None], e[2]], axis=1).tobytes()\n"
        "    assert ctype == e[0] and zlib.decompress(z) == data and z == U.deflate_host(data)[0]\n"
        "print('groups ok')\n" % U.ROOT)
    env = dict(os.environ, PNGLOSS_HIP_DEFLATE_GROUP_BYTES="300000", PNGLOSS_HIP_DEBUG="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "groups ok" in r.stdout, r.stderr[-800:]
    assert r.stderr.count("pngloss_hip deflate:") >= 4          # the seven images went through several groups


def test_stream_only_leaves_the_host_pixels_alone(ctx):
    a = P.synth_rgba(300, 200, 0, 4)
    outs, filts, streams = ctx.run_host_zlib([a], stream_only=True)
    assert np.array_equal(outs[0], a) and not filts[0].any()           # nothing was copied back ...
    _, _, full = ctx.run_host_zlib([a])
    assert streams[0] == full[0]                                       # ... and the stream is the same


def test_too_small_a_buffer_is_refused(ctx):
    """capacity < what the stream needs: PNGLOSS_INVALID_ARGUMENT (4), nothing written past the buffer"""
    import ctypes as C
    from pngloss_amd import lib as L
    a = np.ascontiguousarray(np.random.default_rng(8).integers(0, 256, (64, 64, 4), dtype=np.uint8))
    filt = np.zeros(64, np.uint8)
    buf = np.full(512, 0xAB, np.uint8)                      # a 64x64 noise image needs ~16 KB
    imgs = (L.HostImage * 1)(L.HostImage(a.ctypes.data, filt.ctypes.data, 64, 64))
    zs = (L.ZStream * 1)(L.ZStream(buf.ctypes.data, 256, 0, -1, (C.c_uint32 * 3)(0, 0, 0), 0))
    res = (L.Result * 1)()
    rc = ctx._lib.pngloss_hip_optimize_batch_host_zlib(ctx._ctx, imgs, 1, 0, 2, res, zs)
    assert rc == 4 and zs[0].size == 0
    assert (buf[256:] == 0xAB).all()
    assert ctx._lib.pngloss_hip_zlib_bound(64, 64) >= 64 * (64 * 4 + 1) + 11
