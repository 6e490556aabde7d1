"""Generates the fixtures of the PNG READ side, in the build container only:

  png_read_cases.npz   "<case>/png"  the bytes of a PNG file written by the little encoder below (every colour type x bit depth the
                                     format allows, with and without tRNS, a random filter type 0..4 on every row, IDAT split in
                                     several chunks, sizes around the 64-row band and the 1024-byte block of the device kernel)
                       "<case>/rgba" what the REAL reference reader makes of it: rwpng_read_image24 (/root/reference/src/rwpng.c:422)
                                     through oracle/_ref/librwpng_ref.so (oracle/ref_read_shim.c)
  suite_png.npz        "<name>"      the bytes of /root/reference/suite/<name>.png (data files of the reference; their decoded RGBA8 is
                                     tests/golden/suite_inputs.npz, checked here against the reference reader once more)

usage: python tests/golden/make_png_read_golden.py"""
import ctypes as C
import glob
import os
import struct
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def filter_rows(raw, bpp, rng, forced=None):
    """raw: (H, rowbytes) uint8 unfiltered scanlines -> bytes of the filtered stream with a filter type byte per row."""
    h, rb = raw.shape
    out = bytearray()
    prev = np.zeros(rb, np.int32)
    for y in range(h):
        cur = raw[y].astype(np.int32)
        ft = int(rng.integers(0, 5)) if forced is None else forced
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if rb > bpp else np.zeros(rb, np.int32)
        ul = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if rb > bpp else np.zeros(rb, np.int32)
        if ft == 0: f = cur
        elif ft == 1: f = cur - left
        elif ft == 2: f = cur - prev
        elif ft == 3: f = cur - ((left + prev) >> 1)
        else: f = cur - np.array([paeth(int(a), int(b), int(c)) for a, b, c in zip(left, prev, ul)], np.int32)
        out.append(ft)
        out += (f & 255).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def make_png(w, h, ctype, depth, trns, rng, forced=None):
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bits = channels * depth
    rowbytes = (w * bits + 7) // 8
    bpp = max(1, bits // 8)
    raw = rng.integers(0, 256, (h, rowbytes), dtype=np.uint8)
    if depth < 8 and (w * bits) % 8:
        raw[:, -1] &= (0xff << (8 - (w * bits) % 8)) & 0xff          # padding bits zero
    extra = b""
    if ctype == 3:
        n = 1 << depth
        pal = rng.integers(0, 256, (n, 3), dtype=np.uint8)
        extra += chunk(b"PLTE", pal.tobytes())
        if trns:
            extra += chunk(b"tRNS", rng.integers(0, 256, max(1, n // 2), dtype=np.uint8).tobytes())
    elif trns and ctype == 0:
        # make the key value really occur
        key = int(rng.integers(0, 1 << depth)) if depth < 16 else int(raw[0, 0]) << 8 | int(raw[0, 1])
        extra += chunk(b"tRNS", struct.pack(">H", key))
        if depth == 8: raw[rng.random((h, rowbytes)) < 0.2] = key
    elif trns and ctype == 2:
        if depth == 8:
            key = [int(v) for v in rng.integers(0, 256, 3)]
            px = raw.reshape(h, w, 3); px[rng.random((h, w)) < 0.25] = key
            extra += chunk(b"tRNS", struct.pack(">HHH", *key))
        else:
            px = raw.reshape(h, w, 6)
            key6 = px[0, 0].copy(); px[rng.random((h, w)) < 0.25] = key6
            extra += chunk(b"tRNS", bytes(key6))
    stream = zlib.compress(filter_rows(raw, bpp, rng, forced), 6)
    cut = max(1, len(stream) // 3)
    idat = b"".join(chunk(b"IDAT", stream[i:i + cut]) for i in range(0, len(stream), cut))
    ihdr = struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + extra + idat + chunk(b"IEND", b"")


def ref_reader():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "librwpng_ref.so"))
    lib.ref_read_rgba.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    lib.ref_read_rgba.restype = C.c_int
    lib.ref_read_free.argtypes = [C.c_void_p]

    def read(png_bytes):
        with tempfile.NamedTemporaryFile(suffix=".png", delete=False) as fh:
            fh.write(png_bytes); path = fh.name
        p, w, h = C.c_void_p(), C.c_uint(), C.c_uint()
        rc = lib.ref_read_rgba(path.encode(), C.byref(p), C.byref(w), C.byref(h))
        os.unlink(path)
        assert rc == 0, rc
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 4)).copy()
        lib.ref_read_free(p)
        return a
    return read


CASES = []
for ctype, depths in [(0, [1, 2, 4, 8, 16]), (2, [8, 16]), (3, [1, 2, 4, 8]), (4, [8, 16]), (6, [8, 16])]:
    for depth in depths:
        for trns in ([False, True] if ctype in (0, 2, 3) else [False]):
            for (w, h) in [(37, 19), (130, 70)]:
                CASES.append((ctype, depth, trns, w, h, None))
CASES += [(6, 8, False, 300, 130, None), (2, 8, True, 1030, 66, None), (0, 1, False, 9000, 3, None), (6, 16, False, 140, 65, None),
          (3, 4, True, 2100, 5, None), (6, 8, False, 1, 1, None), (0, 8, False, 1, 200, None), (2, 8, False, 257, 64, None)]
CASES += [(6, 8, False, 90, 40, ft) for ft in range(5)] + [(2, 16, False, 50, 70, 4), (4, 8, False, 64, 128, 3)]


def case_name(c):
    return "t%d_d%d_%s_%dx%d_%s" % (c[0], c[1], "trns" if c[2] else "plain", c[3], c[4], "mix" if c[5] is None else "f%d" % c[5])


def main():
    read = ref_reader()
    rng = np.random.default_rng(20260929)
    out = {}
    for c in CASES:
        png = make_png(c[3], c[4], c[0], c[1], c[2], rng, c[5])
        out[case_name(c) + "/png"] = np.frombuffer(png, np.uint8)
        out[case_name(c) + "/rgba"] = read(png)
    np.savez_compressed(os.path.join(HERE, "png_read_cases.npz"), **out)
    suite = {}
    inputs = np.load(os.path.join(HERE, "suite_inputs.npz"))
    for f in sorted(glob.glob("/root/reference/suite/*.png")):
        name = os.path.basename(f)[:-4]
        data = open(f, "rb").read()
        assert np.array_equal(read(data), inputs[name]), name
        suite[name] = np.frombuffer(data, np.uint8)
    np.savez(os.path.join(HERE, "suite_png.npz"), **suite)
    print(len(CASES), "generated cases,", len(suite), "suite files")


if __name__ == "__main__":
    main()
