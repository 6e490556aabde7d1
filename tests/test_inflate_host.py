"""The device inflater's body (pngloss_amd/csrc/pl_inflate_core.h: one wave per zlib stream) run on the CPU by tests/c/inflate_host.cpp -- the
same source hipcc compiles, lane loops instead of lanes -- against zlib: every block type, every strategy zlib has, window sizes, matches
that overlap themselves and matches at the full 32 KB distance, streams cut into many blocks, the image data of the 78 PNG fixtures
(67 generated + the reference's suite), and the ways a stream can be wrong."""
import ctypes as C
import os
import subprocess
import tempfile
import zlib

import numpy as np
import pytest

from pngloss_amd import lib as L
from tests import util as U

_lib = None


def inflate_lib():
    global _lib
    if _lib is None:
        so = os.path.join(tempfile.mkdtemp(prefix="inflate_host_"), "libinflate_host.so")
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-o", so, os.path.join(U.ROOT, "tests", "c", "inflate_host.cpp")], check=True)
        lib = C.CDLL(so)
        lib.inflate_host.argtypes = [C.c_char_p, C.c_uint, C.c_void_p, C.c_uint]
        lib.inflate_host.restype = C.c_int
        _lib = lib
    return _lib


def run(z, expect):
    out = np.zeros(max(1, expect), np.uint8)
    rc = inflate_lib().inflate_host(z, len(z), out.ctypes.data, expect)
    return rc, out[:expect].tobytes()


def _z(data, **kw):
    co = zlib.compressobj(**kw)
    return co.compress(data) + co.flush()


def _inputs():
    rng = np.random.default_rng(1)
    smooth = (np.cumsum(rng.integers(-2, 3, 300000)) % 256).astype(np.uint8).tobytes()
    rand = rng.integers(0, 256, 100000, dtype=np.uint8).tobytes()
    rle = bytes([7]) * 200000
    far = rand[:32768] + rand[:32768] + rand[:100]                 # matches at the full window distance
    mixed = smooth[:50000] + rle[:70000] + rand[:30000] + smooth[50000:120000]
    return dict(smooth=smooth, rand=rand, rle=rle, far=far, mixed=mixed, one=b"x", tiny=b"abcabcabcabc")


@pytest.mark.parametrize("name", ["smooth", "rand", "rle", "far", "mixed", "one", "tiny"])
def test_inflater_matches_zlib(name):
    d = _inputs()[name]
    for kw in [dict(level=l) for l in (0, 1, 6, 9)] + [dict(level=9, strategy=s) for s in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED)] + [dict(level=9, wbits=9)]:
        rc, out = run(_z(d, **kw), len(d))
        assert rc == 0 and out == d, (name, kw, rc)


def test_inflater_many_blocks_and_staged_input():
    d = _inputs()["mixed"]
    co = zlib.compressobj(9)
    parts = []
    for i in range(0, len(d), 3000):
        parts += [co.compress(d[i:i + 3000]), co.flush(zlib.Z_FULL_FLUSH)]
    parts.append(co.flush())
    rc, out = run(b"".join(parts), len(d))
    assert rc == 0 and out == d
    big = (_inputs()["rand"] * 3)[:250000]                          # incompressible: the staged input turns over many times
    for lvl in (0, 6):
        rc, out = run(_z(big, level=lvl), len(big))
        assert rc == 0 and out == big


def test_inflater_png_fixtures():
    n = 0
    for name, png, _ in U.png_read_fixtures():
        p = L.parse_png(png)
        rc, out = run(p["zstream"], len(p["scanlines"]))
        assert rc == 0 and out == p["scanlines"], name
        n += 1
    assert n == 78


def test_inflater_refuses_what_is_wrong():
    d = _inputs()["smooth"]
    z = _z(d, level=6)
    assert run(z, len(d) - 1)[0] == 6 and run(z, len(d) + 1)[0] == 6            # more / fewer bytes than the image needs
    assert run(z[:len(z) // 2], len(d))[0] != 0                                  # truncated
    assert run(z[:-1] + bytes([z[-1] ^ 1]), len(d))[0] == 7                      # Adler-32
    assert run(b"\x79" + z[1:], len(d))[0] == 1                                  # header check bits
    assert run(z[:2] + b"\x07" + z[3:], len(d))[0] != 0                          # reserved block type
    assert run(_inputs()["rand"][:5000], 100000)[0] != 0                         # not a stream at all
    co = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_DEFAULT_STRATEGY, d[:1000])
    assert run(co.compress(d) + co.flush(), len(d))[0] == 1                      # preset dictionary


def _far_match_stream(dist, length=258, prefix_len=32768, lit=None):
    """A hand-built zlib stream (zlib itself never emits distances beyond 32506; libdeflate and zopfli do): `prefix_len` random bytes in stored
    blocks, then ONE fixed-Huffman block holding a single match (length 258 = code 285, distance code 29 + 13 extra bits) and the end code."""
    rng = np.random.default_rng(dist)
    if lit is None:
        lit = rng.integers(0, 256, prefix_len, dtype=np.uint8).tobytes()
    out = bytearray(b"\x78\x01")
    for i in range(0, len(lit), 65535):
        part = lit[i:i + 65535]
        out += bytes([0]) + len(part).to_bytes(2, "little") + (len(part) ^ 0xFFFF).to_bytes(2, "little") + part
    bits = []

    def put(v, n, msb_first=False):
        for k in (range(n - 1, -1, -1) if msb_first else range(n)):
            bits.append((v >> k) & 1)
    put(1, 1); put(1, 2)                                   # BFINAL, BTYPE = 01 (fixed codes)
    assert length == 258 and 24577 <= dist <= 32768
    put(0xC0 + (285 - 280), 8, True)                       # length code 285, no extra bits
    put(29, 5, True); put(dist - 24577, 13)                # distance code 29 (24577..32768), extra bits LSB first
    put(0, 7, True)                                        # end of block (256)
    while len(bits) % 8:
        bits.append(0)
    out += bytes(sum(b << k for k, b in enumerate(bits[i:i + 8])) for i in range(0, len(bits), 8))
    data = bytearray(lit)
    for k in range(length):
        data.append(data[len(data) - dist])
    out += zlib.adler32(bytes(data)).to_bytes(4, "big")
    assert zlib.decompress(bytes(out)) == bytes(data)      # (zlib's inflate accepts what its deflate never writes)
    return bytes(out), bytes(data)


@pytest.mark.parametrize("dist", [32768, 32767, 32768 - 100, 32768 - 257, 32768 - 258])
def test_inflater_match_whose_destination_wraps_onto_its_source(dist):
    # dist + len > 32768: in the 32 KB ring the slot a late byte is written to is the slot an earlier byte was read from
    z, want = _far_match_stream(dist)
    rc, out = run(z, len(want))
    assert rc == 0 and out == want


def test_inflater_takes_every_path():
    """The decoder's counters (tests/c/inflate_host.cpp: PLI_STAT): a photograph-like stream goes through rounds of several sets, runs of literals, plain matches and codes longer
    than the direct tables; long matches and matches at the window's end take the general path (the cases above) -- and every one of them equals zlib."""
    lib = inflate_lib()
    lib.inflate_host_stats.argtypes = [C.c_void_p]
    rng = np.random.default_rng(5)
    # a skewed alphabet (codes from 2 to 15 bits) with repeats: literals, matches, long codes
    vals = np.minimum(255, rng.geometric(0.06, 400000)).astype(np.uint8)
    data = bytearray(vals.tobytes())
    for i in range(0, len(data) - 4000, 3000):
        data[i + 1000:i + 1000 + 40] = data[i:i + 40]
    data = bytes(data)
    st = (C.c_ulonglong * 8)()
    lib.inflate_host_stats(st)
    rc, out = run(_z(data, level=6), len(data))
    assert rc == 0 and out == data
    lib.inflate_host_stats(st)
    rounds, sets, lits, matches, mbytes, longs, runs = [int(v) for v in st[:7]]
    assert rounds > 1000 and sets > 2 * rounds and lits > 100000 and matches > 50 and mbytes >= 40 * 50 and longs > 100 and runs > 1000, list(st)
