"""pngloss_amd -- MI355X (gfx950) implementation of pngloss's filter+quantise hot path.

The product is the C-ABI shared library ``pngloss_amd/csrc/libpngloss_hip.so`` (see ``include/pngloss_hip.h``);
this package is only the thin ctypes mirror used by the tests, ``bench.py`` and ``__graft_entry__.py``.  It mirrors
the reference's interface for the path (same function names, argument meaning and return codes as
/root/reference/src/pngloss_image.h:14-29) and adds the device-resident batch entry points.

There is no CPU fallback in this package: if the HIP library is missing or no GPU is present the calls raise.
"""
from .lib import (  # noqa: F401
    PNG_FILTER_FLAGS,
    HipContext,
    HipMulti,
    multi_split,
    source_digest,
    build,
    hip_lib,
    optimize_with_rows,
    optimize_with_stride,
    optimize_for_average_filter,
    optimize_image,
    synth_lib,
)
from .synth import SURVEY_FNV_BASIS, fnv1a64, synth_rgba  # noqa: F401

__all__ = [
    "PNG_FILTER_FLAGS", "HipContext", "HipMulti", "multi_split", "source_digest", "build", "hip_lib", "synth_lib", "optimize_with_rows", "optimize_with_stride",
    "optimize_for_average_filter", "optimize_image", "synth_rgba", "fnv1a64", "SURVEY_FNV_BASIS",
]
