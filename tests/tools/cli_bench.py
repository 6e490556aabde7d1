"""End-to-end command line comparison on the GPU box: the reference tool (one file at a time, single thread) vs
pngloss_amd/cli/pngloss (decode threads -> one GPU batch -> encode threads) on the same PNG files."""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pngloss_amd as P
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "pngloss_ref_cli")
OURS = os.path.join(ROOT, "pngloss_amd", "cli", "pngloss")
n, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(os.environ.get("CLI_BENCH_W", 1280)), int(os.environ.get("CLI_BENCH_H", 720))
with tempfile.TemporaryDirectory() as d:
    files = []
    for i in range(n):
        p = os.path.join(d, f"f{i:03d}.png")
        Image.fromarray(P.synth_rgba(W, H, 0, i), "RGBA").save(p, compress_level=1)
        files.append(p)
    if os.environ.get("CLI_BENCH_GPU_ONLY"):
        # large jobs: only the all-GPU path (several windows of 256 files, decode of the next one overlapped)
        t = time.perf_counter()
        r = subprocess.run([OURS, "-f", "--gpu-deflate", "--ext", "-gpu.png"] + files, capture_output=True, text=True, env=dict(os.environ, PNGLOSS_TIMING="1")); print(r.stderr.strip())
        t_gpu = time.perf_counter() - t
        assert r.returncode == 0, r.stderr[-500:]
        print(f"{n} files {W}x{H} with --gpu-deflate: {t_gpu:.2f} s ({n*W*H/t_gpu/1e6:.1f} Mpx/s end to end)")
        sys.exit(0)
    t = time.perf_counter()
    r = subprocess.run([OURS, "-f", "--ext", "-ours.png"] + files, capture_output=True, text=True, env=dict(os.environ, PNGLOSS_TIMING="1")); print(r.stderr.strip())
    t_ours = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-500:]
    t = time.perf_counter()
    r = subprocess.run([OURS, "-f", "--gpu-deflate", "--ext", "-gpu.png"] + files, capture_output=True, text=True, env=dict(os.environ, PNGLOSS_TIMING="1")); print(r.stderr.strip())
    t_gpu = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-500:]
    size_ours = sum(os.path.getsize(p[:-4] + "-ours.png") for p in files)
    size_gpu = sum(os.path.getsize(p[:-4] + "-gpu.png") for p in files)
    same_px = all(np.array_equal(np.array(Image.open(p[:-4] + "-ours.png")), np.array(Image.open(p[:-4] + "-gpu.png"))) for p in files[:4])
    print(f"{n} files {W}x{H} with --gpu-deflate: {t_gpu:.2f} s ({n*W*H/t_gpu/1e6:.1f} Mpx/s end to end); total size {size_gpu} B vs zlib-9 {size_ours} B "
          f"({size_gpu/size_ours:.4f}); same decoded pixels: {same_px}")
    nref = min(n, 4)
    t = time.perf_counter()
    for p in files[:nref]:
        rr = subprocess.run([REF, "-f", "--ext", "-ref.png", p], capture_output=True)
        assert rr.returncode == 0
    t_ref = (time.perf_counter() - t) / nref * n
    same = all(open(p[:-4] + "-ours.png", "rb").read() == open(p[:-4] + "-ref.png", "rb").read() for p in files[:nref])
    print(f"{n} files {W}x{H}: ours {t_ours:.2f} s ({n*W*H/t_ours/1e6:.1f} Mpx/s end to end), reference tool {t_ref:.1f} s extrapolated from {nref} files "
          f"({n*W*H/t_ref/1e6:.2f} Mpx/s); identical output files: {same}")
