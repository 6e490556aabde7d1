"""PNG READ side: inflated IDAT bytes -> RGBA8.  Expected outputs come from the REAL reference reader
(rwpng_read_image24, /root/reference/src/rwpng.c:422, through oracle/_ref/librwpng_ref.so in the build container,
tests/golden/make_png_read_golden.py): every colour type x bit depth PNG allows, with and without tRNS, random filter types per row,
plus the reference's eleven suite files.
  not gpu: the pixel arithmetic shared with the kernel (pl_pngread_core.h) on the CPU
  gpu:     the device reader through the C ABI (pngloss_hip_png_decode_batch_host), one batch of all 78 files"""
import numpy as np
import pytest

import pngloss_amd as P
from pngloss_amd import lib as L
from tests import util as U


def test_pixel_arithmetic_matches_the_reference_reader():
    lib = U.pngread_host_lib()
    n = 0
    for name, png, want in U.png_read_fixtures():
        p = L.parse_png(png)
        assert not p["interlace"]
        out = np.zeros((p["height"], p["width"], 4), np.uint8)
        rc = lib.pngread_host_decode(p["scanlines"], p["width"], p["height"], p["ctype"], p["depth"], p["plte"], len(p["plte"]) // 3 if p["plte"] else 0,
                                     p["trns"], len(p["trns"]) if p["trns"] else 0, out.ctypes.data)
        assert rc == 0 and np.array_equal(out, want), name
        n += 1
    assert n == 78


def test_four_byte_inverse_filter_matches_the_byte_one():
    """pr_recon4 (the kernel's branch-free step for 4-byte pixels) == pr_recon on every byte, all five filter types"""
    lib = U.pngread_host_lib()
    lib.pngread_host_recon4_check.restype = __import__("ctypes").c_int
    assert lib.pngread_host_recon4_check(400000, 3) == 0


def test_fixture_set_covers_every_png_format():
    names = [n for n, _, _ in U.png_read_fixtures()]
    for ctype, depths in [(0, [1, 2, 4, 8, 16]), (2, [8, 16]), (3, [1, 2, 4, 8]), (4, [8, 16]), (6, [8, 16])]:
        for d in depths:
            assert any(n.startswith("t%d_d%d_plain" % (ctype, d)) for n in names)
            if ctype in (0, 2, 3):
                assert any(n.startswith("t%d_d%d_trns" % (ctype, d)) for n in names)


@pytest.mark.gpu
def test_device_reader_matches_the_reference_reader():
    fx = U.png_read_fixtures()
    ctx = P.HipContext()
    outs = ctx.png_decode([png for _, png, _ in fx])
    for (name, _, want), out in zip(fx, outs):
        assert np.array_equal(out, want), (name, np.argwhere((out != want).any(axis=2))[:3].tolist())
    # one at a time too (band / block bookkeeping must not depend on the batch)
    for name, png, want in fx[::9]:
        assert np.array_equal(ctx.png_decode([png])[0], want), name
    ctx.close()


@pytest.mark.gpu
def test_device_reader_rejects_what_is_not_png():
    ctx = P.HipContext()
    good = [f for f in U.png_read_fixtures() if f[0].startswith("t6_d8_plain_37x19")][0][1]
    p = L.parse_png(good)
    bad_rows = bytearray(p["scanlines"]); bad_rows[0] = 7                       # filter type 7
    out = np.zeros((p["height"], p["width"], 4), np.uint8)
    src = (L.PngSource * 1)(L.PngSource(bytes(bad_rows), p["width"], p["height"], p["ctype"], p["depth"], None, 0, None, 0, out.ctypes.data))
    lib = P.hip_lib()
    import ctypes as C
    lib.pngloss_hip_png_decode_batch_host.argtypes = [C.c_void_p, C.POINTER(L.PngSource), C.c_size_t]      # (set here too: the test must not depend on an earlier one having called the wrapper)
    lib.pngloss_hip_png_decode_batch_host.restype = C.c_int
    assert lib.pngloss_hip_png_decode_batch_host(ctx._ctx, src, 1) == 25
    src[0] = L.PngSource(p["scanlines"], p["width"], p["height"], 2, 4, None, 0, None, 0, out.ctypes.data)   # RGB with 4 bits: no such format
    assert lib.pngloss_hip_png_decode_batch_host(ctx._ctx, src, 1) == 4
    ctx.close()


@pytest.mark.gpu
def test_device_reader_one_damaged_file_fails_alone():
    """A window of files with ONE damaged stream (a filter type beyond 4): pngloss_hip_png_decode_batch_host_status reports 25 for that
    file only, the others come back decoded -- the reference's loop, too, fails only the damaged file (pngloss.c:196-204)."""
    fx = [f for f in U.png_read_fixtures() if f[0].startswith("t6_d8") or f[0].startswith("t2_d8")][:4]
    assert len(fx) >= 3
    ctx = P.HipContext()
    good_outs = ctx.png_decode([f[1] for f in fx])
    # damage the second file's first scanline: rebuild a PNG around the altered scanlines is not needed -- patch the parsed bytes
    parsed = [L.parse_png(f[1]) for f in fx]
    rows = bytearray(parsed[1]["scanlines"]); rows[0] = 9
    outs = [np.zeros((p["height"], p["width"], 4), np.uint8) for p in parsed]
    src = (L.PngSource * len(parsed))()
    for i, (p, o) in enumerate(zip(parsed, outs)):
        sc = bytes(rows) if i == 1 else p["scanlines"]
        src[i] = L.PngSource(sc, p["width"], p["height"], p["ctype"], p["depth"], p["plte"], len(p["plte"]) // 3 if p["plte"] else 0, p["trns"], len(p["trns"]) if p["trns"] else 0, o.ctypes.data)
    import ctypes as C
    st = (C.c_int * len(parsed))()
    lib = P.hip_lib()
    lib.pngloss_hip_png_decode_batch_host_status.restype = C.c_int
    lib.pngloss_hip_png_decode_batch_host_status.argtypes = [C.c_void_p, C.POINTER(L.PngSource), C.c_size_t, C.POINTER(C.c_int)]
    rc = lib.pngloss_hip_png_decode_batch_host_status(ctx._ctx, src, len(parsed), st)
    assert rc == 25 and list(st) == [0, 25] + [0] * (len(parsed) - 2)
    for i, (o, g) in enumerate(zip(outs, good_outs)):
        if i != 1:
            assert np.array_equal(o, g), fx[i][0]
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,ctype,depth", [(2000, 700, 6, 8), (4099, 130, 2, 8), (1500, 333, 0, 16), (3000, 200, 4, 8), (5000, 129, 3, 4),
                                             (241, 64, 6, 8), (240, 65, 6, 8), (7, 1000, 2, 16)])
def test_device_reader_many_bands_and_blocks(w, h, ctype, depth):
    """Inverse filtering is defined on any byte stream: random scanlines with random filter types (none, sub, up, average, paeth; mostly
    the last two, which couple a row to the one above), sized so that an image is many bands of 64 rows -- one wave each, handing
    its last row to the band below block by block -- and a row many 960-byte blocks.  Expected: the same pixel arithmetic on the CPU
    (tests/c/pngread_host.cpp, pinned to the reference reader by the fixtures above)."""
    import ctypes as C
    rng = np.random.default_rng(w * 131 + h)
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    rowbytes = (w * channels * depth + 7) // 8
    rows = rng.integers(0, 256, (h, 1 + rowbytes), dtype=np.uint8)
    rows[:, 0] = rng.choice([0, 1, 2, 3, 4, 3, 4, 4], h)
    scan = rows.tobytes()
    plte = bytes(rng.integers(0, 256, 3 * 16, dtype=np.uint8)) if ctype == 3 else None
    trns = bytes(rng.integers(0, 256, 9, dtype=np.uint8)) if ctype == 3 else None
    want = np.zeros((h, w, 4), np.uint8)
    assert U.pngread_host_lib().pngread_host_decode(scan, w, h, ctype, depth, plte, 16 if plte else 0, trns, 9 if trns else 0, want.ctypes.data) == 0
    got = np.zeros((h, w, 4), np.uint8)
    src = (L.PngSource * 1)(L.PngSource(scan, w, h, ctype, depth, plte, 16 if plte else 0, trns, 9 if trns else 0, got.ctypes.data))
    ctx = P.HipContext()
    lib = P.hip_lib()
    lib.pngloss_hip_png_decode_batch_host.restype = C.c_int
    lib.pngloss_hip_png_decode_batch_host.argtypes = [C.c_void_p, C.POINTER(L.PngSource), C.c_size_t]
    assert lib.pngloss_hip_png_decode_batch_host(ctx._ctx, src, 1) == 0
    assert np.array_equal(got, want), np.argwhere((got != want).any(axis=2))[:3].tolist()
    ctx.close()


def _device_bytes(ptr, nbytes):
    """nbytes from a raw device pointer (hipMemcpy of the HIP runtime this process has already loaded)."""
    import ctypes as C
    path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
    hip = C.CDLL(path)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    out = np.zeros(nbytes, np.uint8)
    assert hip.hipMemcpy(out.ctypes.data, ptr, nbytes, 2) == 0          # hipMemcpyDeviceToHost
    return out


@pytest.mark.gpu
def test_device_frames_go_from_the_reader_to_the_optimiser_without_leaving_the_device():
    """SURVEY 8 f.2, "fuse with K0": pngloss_hip_png_decode_batch_device leaves the decoded RGBA8 of the whole fixture set (67 generated files of
    every colour type / bit depth + the reference's eleven suite files) in the context's frame arena; the frames (i) equal the REAL reference
    reader's output and (ii) are handed, as device pointers, straight to the batched optimiser of the same context, whose result equals the
    oracle's on the reference reader's pixels.  The inflated scanlines go up from page-locked memory (pngloss_hip_pinned_alloc)."""
    import torch
    fx = U.png_read_fixtures()
    ctx = P.HipContext()
    frames, st = ctx.png_decode_device([png for _, png, _ in fx])
    assert len(frames) == 78 and not any(st)
    for (name, _, want), (ptr, w, h) in zip(fx, frames):
        assert (h, w) == want.shape[:2] and ptr % 256 == 0, name
        assert np.array_equal(_device_bytes(ptr, w * h * 4).reshape(h, w, 4), want), name
    # the suite files (and a few generated ones): optimise the frames where they are
    pick = [i for i, f in enumerate(fx) if f[0].startswith("suite_")] + list(range(0, 67, 11))
    filt = [torch.zeros(fx[i][2].shape[0], dtype=torch.uint8, device="cuda") for i in pick]
    res = ctx.run([(frames[i][0], f.data_ptr(), frames[i][1], frames[i][2]) for i, f in zip(pick, filt)], 19, 2)
    torch.cuda.synchronize()
    for i, f, r in zip(pick, filt, res):
        want, wf = U.run_port(fx[i][2], 19, 2)
        ptr, w, h = frames[i]
        assert r["status"] == 0, fx[i][0]
        assert np.array_equal(_device_bytes(ptr, w * h * 4).reshape(h, w, 4), want), fx[i][0]
        assert np.array_equal(f.cpu().numpy(), wf), fx[i][0]
    # pageable scanlines work too (the runtime stages them); a second decode reuses the arena
    frames2, st2 = ctx.png_decode_device([fx[3][1], fx[70][1]], pinned=False)
    assert not any(st2)
    for (ptr, w, h), k in zip(frames2, (3, 70)):
        assert np.array_equal(_device_bytes(ptr, w * h * 4).reshape(h, w, 4), fx[k][2])
    ctx.close()


@pytest.mark.gpu
def test_device_inflate_from_file_bytes_to_the_reference_readers_pixels():
    """SURVEY 8 f.2 in full: from the FILE'S COMPRESSED BYTES to RGBA8 on the device -- inflate (one wave per file, pl_inflate_core.h), inverse
    filters and expansion -- for the 67 generated files and the reference's eleven suite files in one batch; the frames equal the REAL
    reference reader's output and feed the optimiser where they are.  A damaged stream fails alone (status 25: that file goes to the host)."""
    import torch
    import zlib
    fx = U.png_read_fixtures()
    ctx = P.HipContext()
    frames, st, rc = ctx.png_decode_device_z([png for _, png, _ in fx])
    assert rc == 0 and not any(st) and len(frames) == 78
    for (name, _, want), (ptr, w, h) in zip(fx, frames):
        assert np.array_equal(_device_bytes(ptr, w * h * 4).reshape(h, w, 4), want), name
    pick = [i for i, f in enumerate(fx) if f[0] in ("suite_rose", "suite_david", "suite_tux")]
    filt = [torch.zeros(fx[i][2].shape[0], dtype=torch.uint8, device="cuda") for i in pick]
    res = ctx.run([(frames[i][0], f.data_ptr(), frames[i][1], frames[i][2]) for i, f in zip(pick, filt)], 19, 2)
    torch.cuda.synchronize()
    for i, f, r in zip(pick, filt, res):
        want, wf = U.run_port(fx[i][2], 19, 2)
        ptr, w, h = frames[i]
        assert r["status"] == 0 and np.array_equal(_device_bytes(ptr, w * h * 4).reshape(h, w, 4), want) and np.array_equal(f.cpu().numpy(), wf), fx[i][0]
    # other encoders' streams of the same scanlines: stored blocks, fixed codes, Huffman only, small windows, many blocks
    sub = [fx[k] for k in (2, 40, 70, 75)]
    for kw in (dict(level=0), dict(level=9, strategy=zlib.Z_FIXED), dict(level=9, strategy=zlib.Z_HUFFMAN_ONLY), dict(level=1), dict(level=9, wbits=9)):
        zs = []
        for _, png, _ in sub:
            co = zlib.compressobj(**kw)
            zs.append(co.compress(L.parse_png(png)["scanlines"]) + co.flush())
        fr, st, rc = ctx.png_decode_device_z([png for _, png, _ in sub], zstreams=zs)
        assert rc == 0 and not any(st), kw
        for (name, _, want), (ptr, w, h) in zip(sub, fr):
            assert np.array_equal(_device_bytes(ptr, w * h * 4).reshape(h, w, 4), want), (name, kw)
    # one damaged stream among good ones
    good = [f[1] for f in fx[:6]]
    zs = [L.parse_png(g)["zstream"] for g in good]
    zs[2] = zs[2][:len(zs[2]) // 2] + bytes(len(zs[2]) - len(zs[2]) // 2)          # second half zeroed
    zs[4] = zs[4][:-1] + bytes([zs[4][-1] ^ 0x55])                                    # wrong Adler-32
    fr, st, rc = ctx.png_decode_device_z(good, zstreams=zs)
    assert rc == 25 and st[2] == 25 and st[4] == 25 and not any(st[k] for k in (0, 1, 3, 5))
    for k in (0, 1, 3, 5):
        ptr, w, h = fr[k]
        assert np.array_equal(_device_bytes(ptr, w * h * 4).reshape(h, w, 4), fx[k][2])
    ctx.close()


@pytest.mark.gpu
def test_device_inflate_match_whose_destination_wraps_onto_its_source():
    """Matches with dist + len > 32768 (libdeflate / zopfli emit them, zlib's deflate never does): in the 32 KB ring the slot a late byte of the match
    is written to is the slot an earlier byte was read from.  A 336x98 gray image whose 33026 scanline bytes are 32768 literal bytes + one such match
    (hand-built streams, tests/test_inflate_host.py::_far_match_stream), inflated and unfiltered on the device."""
    import struct
    import zlib
    from tests.test_inflate_host import _far_match_stream
    w, h = 336, 98
    assert (w + 1) * h == 32768 + 258
    rng = np.random.default_rng(5)
    lit = bytearray(rng.integers(0, 256, 32768, dtype=np.uint8).tobytes())
    for r in range(0, 32768, w + 1):
        lit[r] = 0                                              # filter type byte of every row: none

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    ctx = P.HipContext()
    for dist in (32768, 32767, 32768 - 100, 32768 - 258):
        z, scan = _far_match_stream(dist, lit=bytes(lit))
        assert all(scan[r] == 0 for r in range(0, len(scan), w + 1))
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + chunk(b"IDAT", z) + chunk(b"IEND", b"")
        fr, st, rc = ctx.png_decode_device_z([png])
        assert rc == 0 and st == [0], dist
        g = np.frombuffer(scan, np.uint8).reshape(h, w + 1)[:, 1:]
        want = np.stack([g, g, g, np.full_like(g, 255)], axis=2)
        assert np.array_equal(_device_bytes(fr[0][0], w * h * 4).reshape(h, w, 4), want), dist
    ctx.close()


@pytest.mark.gpu
def test_device_inflate_more_streams_than_the_device_holds():
    """900 streams in one call: three a CU are in flight (768 on 256 CUs), the others follow as workgroups end.  The 78 fixture files over and over, their scanlines compressed
    again at levels 1 / 6 / 9 in turn (other block structures, other code sets), every frame against the real reference reader's pixels."""
    import zlib
    fx = U.png_read_fixtures()
    n = 900
    files, zs, want = [], [], []
    cache = {}
    for i in range(n):
        k, lvl = i % len(fx), (1, 6, 9)[(i // len(fx)) % 3]
        if (k, lvl) not in cache:
            cache[(k, lvl)] = zlib.compress(L.parse_png(fx[k][1])["scanlines"], lvl)
        files.append(fx[k][1]); zs.append(cache[(k, lvl)]); want.append(fx[k][2])
    ctx = P.HipContext()
    fr, st, rc = ctx.png_decode_device_z(files, zstreams=zs)
    assert rc == 0 and not any(st) and len(fr) == n
    for i, ((ptr, w, h), wn) in enumerate(zip(fr, want)):
        assert np.array_equal(_device_bytes(ptr, w * h * 4).reshape(h, w, 4), wn), (i, fx[i % len(fx)][0])
    ctx.close()
