#!/usr/bin/env python3
"""Workload for profiling the GPU deflate stage: N synthetic 1080p frames (mode M) through the _zlib batch call, twice."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pngloss_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = P.HipContext(0)
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
arrays = [P.synth_rgba(W, H, mode, f) for f in range(n)]
for rep in range(2):
    import time
    t0 = time.perf_counter()
    outs, filts, streams = ctx.run_host_zlib(arrays, stream_only=True)
    print(f"rep {rep}: host call {1e3 * (time.perf_counter() - t0):.0f} ms, engine {ctx.engine_ms:.0f} ms, pipeline {ctx.total_ms:.0f} ms")
    print(f"rep {rep}: deflate stage {ctx.deflate_ms:.1f} ms for {n} frames, {sum(len(z) for _, z, _ in streams)} bytes", flush=True)
