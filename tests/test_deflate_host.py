"""The deflate encoder of the PNG write side (pngloss_amd/csrc/pl_deflate_core.h), run on the CPU by
tests/c/deflate_host.cpp: every stream must inflate (zlib) to its input; sizes are compared with zlib level 9 /
Z_FILTERED, which is what libpng gives the reference (rwpng.c:477-637)."""
import zlib

import numpy as np
import pytest

from tests import util as U



@pytest.fixture(scope="module")
def host():
    return U.deflate_host_lib()


def deflate(lib, data, max_chain=64, min_len=6, block_bytes=262144):
    return U.deflate_host(data, max_chain, min_len, block_bytes)


zlib9f = U.zlib9_filtered


def scanline_stream(rgba, s=19, b=2):
    out, flags = U.run_port(rgba, s, b, True)
    _, ids, rows = U.png_scanlines_reference(out, flags)
    return np.concatenate([ids[:, None], rows], axis=1).tobytes()


def test_roundtrip_short_and_degenerate(host):
    rng = np.random.default_rng(1)
    cases = [b"", b"a", b"ab", b"abc", b"aaaaaa", b"abcabcabcabcabcabc", bytes(300), bytes(range(256)) * 3]
    cases += [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (1, 5, 6, 7, 100, 1000)]
    cases += [bytes([1, 2, 3, 4, 5, 6, 7]) * k for k in (1, 2, 37, 38, 40)]            # runs around the 258 limit
    cases += [bytes(257), bytes(258), bytes(259), bytes(258 * 3 + 1)]
    for data in cases:
        z, _ = deflate(host, data, block_bytes=4096)
        assert zlib.decompress(z) == data


def test_roundtrip_block_boundaries_and_kinds(host):
    rng = np.random.default_rng(2)
    noise = rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes()
    z, stats = deflate(host, noise, block_bytes=70_000)              # incompressible: stored blocks, > 65535 per block
    assert zlib.decompress(z) == noise and stats[0] == 3
    assert len(z) <= len(noise) + 2 + 6 + 3 * 2 * 5
    text = (b"the quick brown fox jumps over the lazy dog. " * 4000)
    for block in (100, 777, 4096, 65536, 262144):
        z, stats = deflate(host, text, block_bytes=block)
        assert zlib.decompress(z) == text
    tiny, stats = deflate(host, b"hello hello hello hello", block_bytes=4096)
    assert stats[1] == 1                                              # a tiny block is cheapest with the fixed codes
    mixed = noise[:50_000] + bytes(50_000) + text[:50_000]
    z, stats = deflate(host, mixed, block_bytes=50_000)
    assert zlib.decompress(z) == mixed and stats[0] >= 1 and stats[2] >= 1


def test_code_length_limit(host):
    # Fibonacci-distributed symbol counts make the unrestricted Huffman tree deeper than 15: exercises the Kraft repair
    fib = [1, 1]
    while len(fib) < 28:
        fib.append(fib[-1] + fib[-2])
    rng = np.random.default_rng(3)
    syms = np.concatenate([np.full(c, i * 7 + 1, np.uint8) for i, c in enumerate(fib)])
    rng.shuffle(syms)
    data = syms.tobytes()
    z, stats = deflate(host, data, block_bytes=1 << 20)
    assert zlib.decompress(z) == data and stats[2] >= 1
    z2, _ = deflate(host, data, min_len=6, block_bytes=1 << 22)
    assert zlib.decompress(z2) == data


def test_window_and_image_limits(host):
    rng = np.random.default_rng(4)
    chunk = rng.integers(0, 256, 500, dtype=np.uint8).tobytes()
    # the same 500 bytes again just inside and just outside the 32 KiB window
    for gap in (32768 - 500 - 1, 32768 - 500, 32768 - 500 + 1, 40000):
        data = chunk + rng.integers(0, 4, gap, dtype=np.uint8).tobytes() + chunk
        z, _ = deflate(host, data)
        assert zlib.decompress(z) == data


@pytest.mark.parametrize("mode", range(6))
def test_scanline_streams_roundtrip_and_size(host, mode):
    import pngloss_amd as P
    data = scanline_stream(P.synth_rgba(160, 120, mode, 0))
    z, _ = deflate(host, data)
    assert zlib.decompress(z) == data
    assert len(z) <= 1.03 * len(zlib9f(data)) + 16                    # small images: within 3 % of zlib level 9


def test_size_vs_zlib9_on_a_larger_frame(host):
    import pngloss_amd as P
    data = scanline_stream(P.synth_rgba(512, 384, 0, 0))
    z, _ = deflate(host, data)
    assert zlib.decompress(z) == data
    assert len(z) <= 0.95 * len(zlib9f(data))                         # measured 0.93: the optimal parse at work


def test_team_encoder_is_byte_identical_to_the_one_thread_encoder(host):
    """pl_deflate_coop.h (the workgroup version: speculative chunk parse + merge, rank sort, OR-ed bit runs) against the
    plain statement of the algorithm, for team sizes that give chunks of 32 positions up to the whole block."""
    import pngloss_amd as P
    rng = np.random.default_rng(6)
    cases = [scanline_stream(P.synth_rgba(200, 120, m, 1)) for m in (0, 1, 2, 5)]
    cases += [b"", b"x", bytes(1000), rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(),
              (b"abcdefgh" * 50 + rng.integers(0, 3, 300, dtype=np.uint8).tobytes()) * 40]
    for data in cases:
        for block in (1500, 262144):
            want, wstats = U.deflate_host(data, block_bytes=block, team=0)
            assert zlib.decompress(want) == data
            for team in (1, 3, 16, 64):
                got, gstats = U.deflate_host(data, block_bytes=block, team=team)
                assert got == want, (len(data), block, team)
                assert np.array_equal(gstats, wstats)
