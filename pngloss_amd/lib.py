"""ctypes mirror of include/pngloss_hip.h (the drop-in seam of /root/reference/src/pngloss_image.h:14-29)."""
import ctypes as C
import os
import subprocess
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_ROOT = os.path.dirname(_HERE)

#: libpng filter flag values written to row_filters (pngloss_image.c:290-306): none, sub, up, average, paeth
PNG_FILTER_FLAGS = (0x08, 0x10, 0x20, 0x40, 0x80)

PNGLOSS_SUCCESS = 0
PNGLOSS_INVALID_ARGUMENT = 4
PNGLOSS_OUT_OF_MEMORY_ERROR = 17
PNGLOSS_HIP_ERROR = 64
PNGLOSS_INTERNAL_ABORT = 65

_lock = threading.Lock()
_hip = None
_synth = None


class PnglossImage(C.Structure):
    _fields_ = [("rows", C.POINTER(C.c_void_p)), ("width", C.c_uint32), ("height", C.c_uint32),
                ("bytes_per_pixel", C.c_uint8)]


class ImageDesc(C.Structure):
    _fields_ = [("d_rgba", C.c_void_p), ("d_row_filters", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class HostImage(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("row_filters", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class Scanlines(C.Structure):
    _fields_ = [("filter_types", C.c_void_p), ("scanlines", C.c_void_p), ("pitch", C.c_size_t), ("color_type", C.c_int)]


class ZStream(C.Structure):
    _fields_ = [("data", C.c_void_p), ("capacity", C.c_size_t), ("size", C.c_size_t), ("color_type", C.c_int),
                ("blocks", C.c_uint32 * 3), ("flags", C.c_uint32)]


class PngSource(C.Structure):
    _fields_ = [("scanlines", C.c_char_p), ("width", C.c_uint32), ("height", C.c_uint32), ("color_type", C.c_uint8), ("bit_depth", C.c_uint8),
                ("palette", C.c_char_p), ("palette_entries", C.c_uint32), ("trns", C.c_char_p), ("trns_bytes", C.c_uint32), ("rgba", C.c_void_p)]


def parse_png(data):
    """Chunk walk + inflate of a PNG file (what pngloss_amd/cli/png_stream_reader.c does in C with zlib): dict(width, height, depth, ctype,
    interlace, plte, trns, zstream, scanlines)."""
    import struct
    import zlib
    data = bytes(data)
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    o, out, idat = 8, dict(plte=None, trns=None), []
    while o + 8 <= len(data):
        n, tag = struct.unpack(">I4s", data[o:o + 8])
        body = data[o + 8:o + 8 + n]
        o += 12 + n
        if tag == b"IHDR":
            out["width"], out["height"], out["depth"], out["ctype"], _, _, out["interlace"] = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            out["plte"] = body
        elif tag == b"tRNS":
            out["trns"] = body
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    out["zstream"] = b"".join(idat)
    out["scanlines"] = zlib.decompress(out["zstream"])
    return out


class PngZSource(C.Structure):
    _fields_ = [("zstream", C.c_char_p), ("zbytes", C.c_size_t), ("width", C.c_uint32), ("height", C.c_uint32), ("color_type", C.c_uint8), ("bit_depth", C.c_uint8),
                ("palette", C.c_char_p), ("palette_entries", C.c_uint32), ("trns", C.c_char_p), ("trns_bytes", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("bytes_per_pixel", C.c_uint32), ("unique_symbols", C.c_uint32),
                ("retried_rows", C.c_uint32), ("repaired_pixels", C.c_uint32)]


def build(verbose: bool = False) -> None:
    """Compile libpngloss_hip.so (hipcc, gfx950) and libpngloss_synth.so (gcc) in-tree."""
    r = subprocess.run(["make", "-C", _CSRC, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("building pngloss_amd/csrc failed")


def _load(name):
    path = os.path.join(_CSRC, name)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the HIP path)")
    return C.CDLL(path)


def synth_lib():
    global _synth
    with _lock:
        if _synth is None:
            lib = _load("libpngloss_synth.so")
            lib.pngloss_synth_rgba.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64]
            lib.pngloss_synth_rgba.restype = None
            lib.pngloss_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
            lib.pngloss_fnv1a64.restype = C.c_uint64
            lib.pngloss_fnv1a64_seed.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
            lib.pngloss_fnv1a64_seed.restype = C.c_uint64
            _synth = lib
        return _synth


#: every symbol include/pngloss_hip.h declares (tests/test_abi.py checks the .so exports each of them)
ABI_SYMBOLS = (
    "optimize_with_rows", "optimize_with_stride", "optimizeForAverageFilter", "optimize_image",
    "pngloss_hip_device_count", "pngloss_hip_create", "pngloss_hip_destroy", "pngloss_hip_optimize_batch_async",
    "pngloss_hip_finish", "pngloss_hip_optimize_batch", "pngloss_hip_optimize_batch_host", "pngloss_hip_optimize_batch_host_emit",
    "pngloss_hip_optimize_batch_host_zlib", "pngloss_hip_zlib_bound", "pngloss_hip_last_deflate_ms", "pngloss_hip_last_engine_ms", "pngloss_hip_last_total_ms",
    "pngloss_hip_last_histogram", "pngloss_hip_last_engine_info", "pngloss_hip_png_decode_batch_host", "pngloss_hip_png_decode_batch_host_status", "pngloss_hip_png_decode_batch_device", "pngloss_hip_png_decode_batch_device_z", "pngloss_hip_pinned_alloc", "pngloss_hip_pinned_free", "pngloss_hip_set_option", "pngloss_hip_version",
    "pngloss_hip_multi_create", "pngloss_hip_multi_destroy", "pngloss_hip_multi_count", "pngloss_hip_multi_split",
    "pngloss_hip_multi_optimize_batch_host",
)


def hip_lib():
    global _hip
    with _lock:
        if _hip is None:
            # PyTorch-ROCm wheels bundle their own libamdhip64; whichever copy is loaded first serves the whole process.
            # Loading /opt/rocm's copy first (through our .so) and torch's afterwards leaves torch without devices, so
            # when torch is installed let it load first.  The C library itself has no torch dependency.
            if "torch" not in sys.modules and os.environ.get("PNGLOSS_NO_TORCH_PRELOAD") is None:
                try:
                    import torch  # noqa: F401
                except ImportError:
                    pass
            lib = _load(os.environ.get("PNGLOSS_HIP_LIBNAME", "libpngloss_hip.so"))
            rows_t = C.POINTER(C.c_void_p)
            lib.optimize_with_rows.argtypes = [rows_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_bool, C.c_uint8, C.c_long]
            lib.optimize_with_rows.restype = C.c_int
            lib.optimize_with_stride.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_bool, C.c_uint8, C.c_long]
            lib.optimize_with_stride.restype = None
            lib.optimizeForAverageFilter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
            lib.optimizeForAverageFilter.restype = None
            lib.optimize_image.argtypes = [C.POINTER(PnglossImage), C.c_void_p, C.c_bool, C.c_uint8, C.c_long]
            lib.optimize_image.restype = C.c_int
            lib.pngloss_hip_device_count.restype = C.c_int
            lib.pngloss_hip_create.argtypes = [C.c_int]
            lib.pngloss_hip_create.restype = C.c_void_p
            lib.pngloss_hip_destroy.argtypes = [C.c_void_p]
            lib.pngloss_hip_destroy.restype = None
            lib.pngloss_hip_optimize_batch_async.argtypes = [C.c_void_p, C.POINTER(ImageDesc), C.c_size_t, C.c_uint, C.c_long, C.c_void_p]
            lib.pngloss_hip_optimize_batch_async.restype = C.c_int
            lib.pngloss_hip_finish.argtypes = [C.c_void_p, C.POINTER(Result), C.c_size_t]
            lib.pngloss_hip_finish.restype = C.c_int
            lib.pngloss_hip_optimize_batch.argtypes = [C.c_void_p, C.POINTER(ImageDesc), C.c_size_t, C.c_uint, C.c_long, C.c_void_p, C.POINTER(Result)]
            lib.pngloss_hip_optimize_batch.restype = C.c_int
            lib.pngloss_hip_optimize_batch_host.argtypes = [C.c_void_p, C.POINTER(HostImage), C.c_size_t, C.c_uint, C.c_long, C.POINTER(Result)]
            lib.pngloss_hip_optimize_batch_host.restype = C.c_int
            lib.pngloss_hip_optimize_batch_host_emit.argtypes = [C.c_void_p, C.POINTER(HostImage), C.c_size_t, C.c_uint, C.c_long, C.POINTER(Result), C.POINTER(Scanlines)]
            lib.pngloss_hip_optimize_batch_host_emit.restype = C.c_int
            lib.pngloss_hip_multi_create.argtypes = [C.c_char_p]
            lib.pngloss_hip_multi_create.restype = C.c_void_p
            lib.pngloss_hip_multi_destroy.argtypes = [C.c_void_p]
            lib.pngloss_hip_multi_destroy.restype = None
            lib.pngloss_hip_multi_count.argtypes = [C.c_void_p]
            lib.pngloss_hip_multi_count.restype = C.c_int
            lib.pngloss_hip_multi_split.argtypes = [C.POINTER(HostImage), C.c_size_t, C.c_int, C.POINTER(C.c_int)]
            lib.pngloss_hip_multi_split.restype = None
            lib.pngloss_hip_multi_optimize_batch_host.argtypes = [C.c_void_p, C.POINTER(HostImage), C.c_size_t, C.c_uint, C.c_long, C.POINTER(Result), C.POINTER(Scanlines), C.POINTER(ZStream)]
            lib.pngloss_hip_multi_optimize_batch_host.restype = C.c_int
            lib.pngloss_hip_optimize_batch_host_zlib.argtypes = [C.c_void_p, C.POINTER(HostImage), C.c_size_t, C.c_uint, C.c_long, C.POINTER(Result), C.POINTER(ZStream)]
            lib.pngloss_hip_optimize_batch_host_zlib.restype = C.c_int
            lib.pngloss_hip_zlib_bound.argtypes = [C.c_uint32, C.c_uint32]
            lib.pngloss_hip_zlib_bound.restype = C.c_size_t
            lib.pngloss_hip_last_deflate_ms.argtypes = [C.c_void_p]
            lib.pngloss_hip_last_deflate_ms.restype = C.c_double
            lib.pngloss_hip_last_engine_ms.argtypes = [C.c_void_p]
            lib.pngloss_hip_last_engine_ms.restype = C.c_double
            lib.pngloss_hip_last_total_ms.argtypes = [C.c_void_p]
            lib.pngloss_hip_last_total_ms.restype = C.c_double
            lib.pngloss_hip_last_histogram.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
            lib.pngloss_hip_last_histogram.restype = C.c_int
            lib.pngloss_hip_version.restype = C.c_char_p
            _hip = lib
        return _hip


PNGLOSS_INTERNAL_ABORT = 65


def _check(rc, what, partial_ok=False):
    """partial_ok: the batch entry points return PNGLOSS_INTERNAL_ABORT (65) when SINGLE images hit the row the reference abort()s on
    (pngloss_image.c:268-271) -- every other image of the batch is done and results[i].status tells which failed, so the wrappers
    hand the per-image statuses back instead of throwing the whole batch away."""
    if rc != PNGLOSS_SUCCESS and not (partial_ok and rc == PNGLOSS_INTERNAL_ABORT):
        raise RuntimeError(f"{what} failed with pngloss_error {rc}")


def _row_pointers(arr: np.ndarray):
    h = arr.shape[0]
    stride = arr.strides[0]
    base = arr.ctypes.data
    return (C.c_void_p * h)(*[base + y * stride for y in range(h)])


# ---- host-pointer seam, same names as the reference ---------------------------------------------------------

def optimize_with_rows(rgba: np.ndarray, strength: int = 19, bleed: int = 2, want_filters: bool = True, verbose: bool = False):
    """optimize_with_rows (pngloss_image.h:21-25) on an (H, W, 4) uint8 array.  Returns (rgba_out, row_filters|None)."""
    assert rgba.dtype == np.uint8 and rgba.ndim == 3 and rgba.shape[2] == 4
    out = np.ascontiguousarray(rgba).copy()
    h, w = out.shape[:2]
    filt = np.zeros(h, np.uint8) if want_filters else None
    rows = _row_pointers(out) if h else (C.c_void_p * 1)()
    rc = hip_lib().optimize_with_rows(rows, w, h, filt.ctypes.data_as(C.c_void_p) if want_filters else None, verbose, strength, bleed)
    _check(rc, "optimize_with_rows")
    return out, filt


def optimize_with_stride(rgba: np.ndarray, strength: int = 19, bleed: int = 2):
    """optimize_with_stride (pngloss_image.h:17-20): contiguous buffer + stride, row_filters = NULL mode."""
    out = np.ascontiguousarray(rgba).copy()
    h, w = out.shape[:2]
    hip_lib().optimize_with_stride(out.ctypes.data_as(C.c_void_p), w, h, out.strides[0], False, strength, bleed)
    return out


def optimize_for_average_filter(rgba: np.ndarray, strength: int):
    """optimizeForAverageFilter (pngloss_image.h:14-16)."""
    out = np.ascontiguousarray(rgba).copy()
    h, w = out.shape[:2]
    hip_lib().optimizeForAverageFilter(out.ctypes.data_as(C.c_void_p), w, h, strength)
    return out


def optimize_image(packed: np.ndarray, strength: int = 19, bleed: int = 2, want_filters: bool = True):
    """optimize_image (pngloss_image.h:26-29) on an (H, W, bpp) packed uint8 array, bpp 1..4."""
    assert packed.dtype == np.uint8 and packed.ndim == 3 and 1 <= packed.shape[2] <= 4
    out = np.ascontiguousarray(packed).copy()
    h, w, bpp = out.shape
    filt = np.zeros(h, np.uint8) if want_filters else None
    rows = _row_pointers(out) if h else (C.c_void_p * 1)()
    img = PnglossImage(C.cast(rows, C.POINTER(C.c_void_p)), w, h, bpp)
    rc = hip_lib().optimize_image(C.byref(img), filt.ctypes.data_as(C.c_void_p) if want_filters else None, False, strength, bleed)
    _check(rc, "optimize_image")
    return out, filt


# ---- device-resident batch extension ------------------------------------------------------------------------

class HipContext:
    """pngloss_hip_ctx wrapper: device-resident batches (pointers come from torch tensors / hipMalloc)."""

    def __init__(self, device: int = -1):
        self._lib = hip_lib()
        self._ctx = self._lib.pngloss_hip_create(device)
        if not self._ctx:
            raise RuntimeError("pngloss_hip_create failed: no usable HIP device (there is no CPU fallback)")

    def close(self):
        if self._ctx:
            self._lib.pngloss_hip_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        """pngloss_hip_set_option: e.g. ("engine", "seg" | "wg" | "auto")"""
        self._lib.pngloss_hip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        self._lib.pngloss_hip_set_option.restype = C.c_int
        _check(self._lib.pngloss_hip_set_option(self._ctx, name.encode(), value.encode()), "set_option")

    def enqueue(self, images, strength=19, bleed=2, stream=0):
        """images: sequence of (d_rgba_ptr, d_filters_ptr_or_0, width, height). Asynchronous."""
        n = len(images)
        descs = (ImageDesc * max(n, 1))()
        for i, (p, f, w, h) in enumerate(images):
            descs[i] = ImageDesc(p, f or None, w, h)
        _check(self._lib.pngloss_hip_optimize_batch_async(self._ctx, descs, n, strength, bleed, stream or None), "enqueue")
        self._n = n

    def finish(self):
        n = getattr(self, "_n", 0)
        res = (Result * max(n, 1))()
        _check(self._lib.pngloss_hip_finish(self._ctx, res, n), "finish", partial_ok=True)
        return [dict(status=r.status, bpp=r.bytes_per_pixel, unique_symbols=r.unique_symbols, retried_rows=r.retried_rows, repaired_pixels=r.repaired_pixels) for r in res[:n]]

    def run(self, images, strength=19, bleed=2, stream=0):
        """pngloss_hip_optimize_batch: the SYNCHRONOUS entry point (enqueue + finish in one call; the library then spares the caller's stream the device-side
        wait for the engine that the asynchronous entry needs)."""
        n = len(images)
        descs = (ImageDesc * max(n, 1))()
        for i, (p, f, w, h) in enumerate(images):
            descs[i] = ImageDesc(p, f or None, w, h)
        res = (Result * max(n, 1))()
        self._n = n
        _check(self._lib.pngloss_hip_optimize_batch(self._ctx, descs, n, strength, bleed, stream or None, res), "optimize_batch", partial_ok=True)
        return [dict(status=r.status, bpp=r.bytes_per_pixel, unique_symbols=r.unique_symbols, retried_rows=r.retried_rows, repaired_pixels=r.repaired_pixels) for r in res[:n]]

    def run_host(self, arrays, strength=19, bleed=2, want_filters=True, inplace=False):
        """pngloss_hip_optimize_batch_host on a list of (H, W, 4) uint8 arrays.  Returns (outs, filters, results).
        inplace: the arrays (C-contiguous uint8) are optimised where they are, like the C call does; otherwise copies are."""
        outs = list(arrays) if inplace else [np.ascontiguousarray(a).copy() for a in arrays]
        if inplace:
            assert all(a.flags["C_CONTIGUOUS"] and a.dtype == np.uint8 for a in outs)
        filts = [np.zeros(a.shape[0], np.uint8) if want_filters else None for a in outs]
        n = len(outs)
        imgs = (HostImage * max(n, 1))()
        for i, (a, f) in enumerate(zip(outs, filts)):
            imgs[i] = HostImage(a.ctypes.data, f.ctypes.data if f is not None else None, a.shape[1], a.shape[0])
        res = (Result * max(n, 1))()
        _check(self._lib.pngloss_hip_optimize_batch_host(self._ctx, imgs, n, strength, bleed, res), "optimize_batch_host", partial_ok=True)
        return outs, filts, [dict(status=r.status, bpp=r.bytes_per_pixel, unique_symbols=r.unique_symbols) for r in res[:n]]

    def run_host_emit(self, arrays, strength=19, bleed=2, want_filters=True):
        """pngloss_hip_optimize_batch_host_emit: like run_host, plus per image (color_type, filter_types[H], scanlines[H, W*ch])."""
        outs = [np.ascontiguousarray(a).copy() for a in arrays]
        filts = [np.zeros(a.shape[0], np.uint8) if want_filters else None for a in outs]
        n = len(outs)
        imgs = (HostImage * max(n, 1))()
        lines = (Scanlines * max(n, 1))()
        ids = [np.zeros(a.shape[0], np.uint8) for a in outs]
        rows = [np.zeros((a.shape[0], a.shape[1] * 4), np.uint8) for a in outs]
        for i, (a, f) in enumerate(zip(outs, filts)):
            imgs[i] = HostImage(a.ctypes.data, f.ctypes.data if f is not None else None, a.shape[1], a.shape[0])
            lines[i] = Scanlines(ids[i].ctypes.data, rows[i].ctypes.data, a.shape[1] * 4, -1)
        res = (Result * max(n, 1))()
        _check(self._lib.pngloss_hip_optimize_batch_host_emit(self._ctx, imgs, n, strength, bleed, res, lines), "optimize_batch_host_emit", partial_ok=True)
        chans = {0: 1, 4: 2, 2: 3, 6: 4}
        emitted = [(lines[i].color_type, ids[i], rows[i][:, : outs[i].shape[1] * chans.get(lines[i].color_type, 4)].copy()) for i in range(n)]
        return outs, filts, emitted

    def run_host_zlib(self, arrays, strength=19, bleed=2, want_filters=True, stream_only=False):
        """pngloss_hip_optimize_batch_host_zlib: like run_host, plus per image (color_type, zlib stream bytes, blocks).
        stream_only: PNGLOSS_HIP_Z_STREAM_ONLY -- the returned pixel arrays are then the unmodified inputs."""
        outs = [np.ascontiguousarray(a).copy() for a in arrays]
        filts = [np.zeros(a.shape[0], np.uint8) if want_filters else None for a in outs]
        n = len(outs)
        imgs = (HostImage * max(n, 1))()
        zs = (ZStream * max(n, 1))()
        bufs = [np.zeros(self._lib.pngloss_hip_zlib_bound(a.shape[1], a.shape[0]), np.uint8) for a in outs]
        for i, (a, f) in enumerate(zip(outs, filts)):
            imgs[i] = HostImage(a.ctypes.data, f.ctypes.data if f is not None else None, a.shape[1], a.shape[0])
            zs[i] = ZStream(bufs[i].ctypes.data, bufs[i].size, 0, -1, (C.c_uint32 * 3)(0, 0, 0), 1 if stream_only else 0)
        res = (Result * max(n, 1))()
        _check(self._lib.pngloss_hip_optimize_batch_host_zlib(self._ctx, imgs, n, strength, bleed, res, zs), "optimize_batch_host_zlib", partial_ok=True)
        streams = [(zs[i].color_type, bufs[i][: zs[i].size].tobytes(), tuple(zs[i].blocks)) for i in range(n)]
        return outs, filts, streams

    @property
    def deflate_ms(self):
        return self._lib.pngloss_hip_last_deflate_ms(self._ctx)

    @property
    def engine_ms(self):
        return self._lib.pngloss_hip_last_engine_ms(self._ctx)

    @property
    def total_ms(self):
        return self._lib.pngloss_hip_last_total_ms(self._ctx)

    def png_decode(self, files):
        """pngloss_hip_png_decode_batch_host on a list of PNG file contents (bytes): chunk parsing and inflate on the host (parse_png), the
        inverse filters and the expansion to RGBA8 on the device.  Returns a list of (H, W, 4) uint8 arrays."""
        parsed = [parse_png(f) for f in files]
        outs = [np.zeros((p["height"], p["width"], 4), np.uint8) for p in parsed]
        src = (PngSource * max(1, len(parsed)))()
        for i, (p, o) in enumerate(zip(parsed, outs)):
            if p["interlace"]:
                raise ValueError("interlaced PNG files are read with libpng, not on the device")
            src[i] = PngSource(p["scanlines"], p["width"], p["height"], p["ctype"], p["depth"], p["plte"], len(p["plte"]) // 3 if p["plte"] else 0,
                               p["trns"], len(p["trns"]) if p["trns"] else 0, o.ctypes.data)
        self._lib.pngloss_hip_png_decode_batch_host.argtypes = [C.c_void_p, C.POINTER(PngSource), C.c_size_t]
        self._lib.pngloss_hip_png_decode_batch_host.restype = C.c_int
        _check(self._lib.pngloss_hip_png_decode_batch_host(self._ctx, src, len(parsed)), "png_decode")
        return outs

    def png_decode_device(self, files, stream=0, pinned=True):
        """pngloss_hip_png_decode_batch_device: the decoded RGBA8 frames STAY on the device.  Returns a list of (device pointer, width, height)
        ready for enqueue()/run() on this context, and the status list.  pinned: the inflated scanlines go up from page-locked memory
        (pngloss_hip_pinned_alloc), as the command line tool does."""
        parsed = [parse_png(f) for f in files]
        n = len(parsed)
        src = (PngSource * max(1, n))()
        self._lib.pngloss_hip_pinned_alloc.argtypes = [C.c_size_t]
        self._lib.pngloss_hip_pinned_alloc.restype = C.c_void_p
        self._lib.pngloss_hip_pinned_free.argtypes = [C.c_void_p]
        self._lib.pngloss_hip_pinned_free.restype = None
        staged = []
        for i, p in enumerate(parsed):
            if p["interlace"]:
                raise ValueError("interlaced PNG files are read with libpng, not on the device")
            sl = p["scanlines"]
            if pinned and len(sl):
                buf = self._lib.pngloss_hip_pinned_alloc(len(sl))
                if not buf:
                    raise RuntimeError("pngloss_hip_pinned_alloc failed")
                C.memmove(buf, sl, len(sl))
                staged.append(buf)
                sl = C.cast(buf, C.c_char_p)
            src[i] = PngSource(sl, p["width"], p["height"], p["ctype"], p["depth"], p["plte"], len(p["plte"]) // 3 if p["plte"] else 0,
                               p["trns"], len(p["trns"]) if p["trns"] else 0, None)
        ptrs = (C.c_void_p * max(1, n))()
        st = (C.c_int * max(1, n))()
        self._lib.pngloss_hip_png_decode_batch_device.argtypes = [C.c_void_p, C.POINTER(PngSource), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]
        self._lib.pngloss_hip_png_decode_batch_device.restype = C.c_int
        try:
            rc = self._lib.pngloss_hip_png_decode_batch_device(self._ctx, src, n, ptrs, st, stream or None)
        finally:
            for b in staged:
                self._lib.pngloss_hip_pinned_free(b)
        if rc not in (0, 25):
            _check(rc, "png_decode_device")
        return [(ptrs[i], parsed[i]["width"], parsed[i]["height"]) for i in range(n)], list(st)[:n]

    def png_decode_device_z(self, files, stream=0, zstreams=None):
        """pngloss_hip_png_decode_batch_device_z: the compressed image data goes up, inflate + inverse filters + expansion run on the device, the
        RGBA8 frames stay there.  Returns ([(device pointer, width, height)], status list, return code).  zstreams: replaces the files' own
        streams (tests hand damaged ones in)."""
        parsed = [parse_png(f) for f in files]
        n = len(parsed)
        src = (PngZSource * max(1, n))()
        keep = []
        for i, p in enumerate(parsed):
            z = zstreams[i] if zstreams is not None else p["zstream"]
            keep.append(z)
            src[i] = PngZSource(z, len(z), p["width"], p["height"], p["ctype"], p["depth"], p["plte"], len(p["plte"]) // 3 if p["plte"] else 0,
                                p["trns"], len(p["trns"]) if p["trns"] else 0)
        ptrs = (C.c_void_p * max(1, n))()
        st = (C.c_int * max(1, n))()
        self._lib.pngloss_hip_png_decode_batch_device_z.argtypes = [C.c_void_p, C.POINTER(PngZSource), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]
        self._lib.pngloss_hip_png_decode_batch_device_z.restype = C.c_int
        rc = self._lib.pngloss_hip_png_decode_batch_device_z(self._ctx, src, n, ptrs, st, stream or None)
        if rc not in (0, 25):
            _check(rc, "png_decode_device_z")
        return [(ptrs[i], parsed[i]["width"], parsed[i]["height"]) for i in range(n)], list(st)[:n], rc

    def png_decode_status(self, files):
        """pngloss_hip_png_decode_batch_host_status: like png_decode, but a damaged file only fails itself.  Returns (outs, status list, return code)."""
        parsed = [parse_png(f) for f in files]
        outs = [np.zeros((p["height"], p["width"], 4), np.uint8) for p in parsed]
        src = (PngSource * max(1, len(parsed)))()
        for i, (p, o) in enumerate(zip(parsed, outs)):
            src[i] = PngSource(p["scanlines"], p["width"], p["height"], p["ctype"], p["depth"], p["plte"], len(p["plte"]) // 3 if p["plte"] else 0,
                               p["trns"], len(p["trns"]) if p["trns"] else 0, o.ctypes.data)
        st = (C.c_int * max(1, len(parsed)))()
        self._lib.pngloss_hip_png_decode_batch_host_status.argtypes = [C.c_void_p, C.POINTER(PngSource), C.c_size_t, C.POINTER(C.c_int)]
        self._lib.pngloss_hip_png_decode_batch_host_status.restype = C.c_int
        rc = self._lib.pngloss_hip_png_decode_batch_host_status(self._ctx, src, len(parsed), st)
        return outs, list(st)[:len(parsed)], rc

    def engine_info(self, index=0):
        """pngloss_hip_last_engine_info: dict(engine, attempts, restarts, serial_rows, none_dropped) for image `index`."""
        a = (C.c_int32 * 8)()
        self._lib.pngloss_hip_last_engine_info.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        self._lib.pngloss_hip_last_engine_info.restype = C.c_int
        _check(self._lib.pngloss_hip_last_engine_info(self._ctx, index, a), "engine_info")
        return dict(engine={3: "segment-parallel", 0: "workgroup-per-image", 4: "row-statistics (strength 0)"}.get(a[0], a[0]), attempts=a[1], restarts=a[2], serial_rows=a[3], none_dropped=a[4], walked_segments=a[5], launch_groups=a[6], stream_wait=a[7])

    def histogram(self, index=0):
        h = np.zeros(256, np.uint32)
        _check(self._lib.pngloss_hip_last_histogram(self._ctx, index, h.ctypes.data_as(C.c_void_p)), "histogram")
        return h


def source_digest():
    """16 hex digits over the text of every kernel / host source of the library (csrc/*.hip, *.h, *.c, sorted by name): what a committed profile was
    taken AT.  tools/gpu_round5.sh stamps every profiles/r05_* file with it and bench.py compares the stamp of the static blocks it quotes with
    the tree it runs from (no .git travels to the GPU box, a content digest does)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".c")):
            h.update(name.encode() + b"\0")
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def multi_split(shapes, parts):
    """pngloss_hip_multi_split (the C host's LPT split) for a list of (width, height); returns owner indices.  No GPU needed."""
    lib = hip_lib()
    n = len(shapes)
    imgs = (HostImage * max(n, 1))()
    for i, (w, h) in enumerate(shapes):
        imgs[i] = HostImage(None, None, w, h)
    owner = (C.c_int * max(n, 1))()
    lib.pngloss_hip_multi_split(imgs, n, parts, owner)
    return list(owner[:n])


class HipMulti:
    """pngloss_hip_multi wrapper: host-memory batches over every GPU of the node (or the devices named, e.g. "0,0")."""

    def __init__(self, devices=None):
        self._lib = hip_lib()
        self._m = self._lib.pngloss_hip_multi_create(devices.encode() if devices else None)
        if not self._m:
            raise RuntimeError("pngloss_hip_multi_create failed: no usable HIP device (there is no CPU fallback)")

    @property
    def count(self):
        return self._lib.pngloss_hip_multi_count(self._m)

    def close(self):
        if self._m:
            self._lib.pngloss_hip_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_host(self, arrays, strength=19, bleed=2, want_filters=True):
        outs = [np.ascontiguousarray(a).copy() for a in arrays]
        filts = [np.zeros(a.shape[0], np.uint8) if want_filters else None for a in outs]
        n = len(outs)
        imgs = (HostImage * max(n, 1))()
        for i, (a, f) in enumerate(zip(outs, filts)):
            imgs[i] = HostImage(a.ctypes.data, f.ctypes.data if f is not None else None, a.shape[1], a.shape[0])
        res = (Result * max(n, 1))()
        _check(self._lib.pngloss_hip_multi_optimize_batch_host(self._m, imgs, n, strength, bleed, res, None, None), "multi_optimize_batch_host", partial_ok=True)
        return outs, filts, [dict(status=r.status, bpp=r.bytes_per_pixel, unique_symbols=r.unique_symbols) for r in res[:n]]
