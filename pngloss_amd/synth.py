"""Deterministic synthetic RGBA8 frames and FNV-1a-64 digests, via libpngloss_synth.so."""
import ctypes as C

import numpy as np

from .lib import synth_lib

#: offset basis SURVEY.md Appendix B's digest table was produced with (decimal FNV basis minus its last digit)
SURVEY_FNV_BASIS = 0x14650FB0739D0383
FNV_BASIS = 0xCBF29CE484222325


def synth_rgba(width: int, height: int, mode: int = 0, frame: int = 0) -> np.ndarray:
    """(height, width, 4) uint8 frame. mode 0 photo (rgba), 1 noise, 2 rgb, 3 gray+alpha, 4 gray, 5 transparent checker."""
    out = np.empty((height, width, 4), np.uint8)
    synth_lib().pngloss_synth_rgba(out.ctypes.data_as(C.c_void_p), width, height, mode, frame)
    return out


def fnv1a64(data, basis: int = FNV_BASIS) -> int:
    a = np.ascontiguousarray(data)
    return int(synth_lib().pngloss_fnv1a64_seed(a.ctypes.data_as(C.c_void_p), a.nbytes, basis))
