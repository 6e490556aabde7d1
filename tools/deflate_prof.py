#!/usr/bin/env python3
"""Workload for profiling the GPU deflate stage: N synthetic 1080p frames (mode M) through the _zlib batch call, twice."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pngloss_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = P.HipContext(0)
arrays = [P.synth_rgba(1920, 1080, mode, f) for f in range(n)]
for rep in range(2):
    outs, filts, streams = ctx.run_host_zlib(arrays)
    print(f"rep {rep}: deflate stage {ctx.deflate_ms:.1f} ms for {n} frames, {sum(len(z) for _, z, _ in streams)} bytes", flush=True)
