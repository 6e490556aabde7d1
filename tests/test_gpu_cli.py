"""GPU drop-in test: the reference's UNMODIFIED command-line program (pngloss.c, pngloss_opts.c, rwpng.c compiled in
place by oracle/Makefile) linked against libpngloss_hip.so instead of its own hot-path sources must write exactly the
files the all-reference build writes -- same pixels, same per-row filter bytes, same PNG bytes (both use the same libpng).
Covers BASELINE.json configs[0] (suite/david.png through the CLI) and the web front-end contract
(`pngloss -sN -bN --strip -`, website/pnglossapi.go:543-556)."""
import os
import subprocess
import zlib

import numpy as np
import pytest

from tests import util as U

pytestmark = pytest.mark.gpu

REF_CLI = os.path.join(U.ROOT, "oracle", "_ref", "pngloss_ref_cli")
HIP_CLI = os.path.join(U.ROOT, "oracle", "_ref", "pngloss_hip_cli")
needs_cli = pytest.mark.skipif(not (os.path.exists(REF_CLI) and os.path.exists(HIP_CLI)),
                               reason="oracle/_ref CLI builds absent (they are built where /root/reference exists)")


def _write_png(path, rgba):
    from PIL import Image
    Image.fromarray(rgba, "RGBA").save(path)


def _idat_filter_bytes(png_bytes, width, height):
    """Filter type byte of every scanline + colour type, straight from the PNG stream."""
    pos, idat, ctype, depth = 8, b"", None, None
    while pos < len(png_bytes):
        n = int.from_bytes(png_bytes[pos:pos + 4], "big")
        kind = png_bytes[pos + 4:pos + 8]
        data = png_bytes[pos + 8:pos + 8 + n]
        if kind == b"IHDR":
            depth, ctype = data[8], data[9]
        if kind == b"IDAT":
            idat += data
        pos += 12 + n
    raw = zlib.decompress(idat)
    channels = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    stride = 1 + width * channels * depth // 8
    return bytes(raw[y * stride] for y in range(height)), ctype


@needs_cli
@pytest.mark.parametrize("name", ["david", "rose", "tux"])
def test_reference_cli_on_our_library_writes_identical_files(tmp_path, name):
    g = U.load_npz("suite_small.npz")
    src = str(tmp_path / f"{name}.png")
    _write_png(src, g[name + "/in"])
    outs = {}
    for tag, exe in (("ref", REF_CLI), ("hip", HIP_CLI)):
        out = str(tmp_path / f"{name}-{tag}.png")
        r = subprocess.run([exe, "-f", "-s", "19", "-b", "2", "-o", out, src], capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-500:]
        outs[tag] = open(out, "rb").read()
    assert outs["hip"] == outs["ref"]
    # and the file really carries the golden pixels and the golden per-row filter choices
    from PIL import Image
    import io
    h, w = g[name + "/in"].shape[:2]
    decoded = np.array(Image.open(io.BytesIO(outs["hip"])).convert("RGBA"))
    assert np.array_equal(decoded, g[name + "/out"])
    filt, _ = _idat_filter_bytes(outs["hip"], w, h)
    want = bytes({0x08: 0, 0x10: 1, 0x20: 2, 0x40: 3, 0x80: 4}[int(v)] for v in g[name + "/filters"])
    assert filt[1:] == want[1:]          # row 0 is written with PNG_ALL_FILTERS by rwpng.c:488-490 ...
    assert filt[0] == want[0]            # ... and libpng's heuristic picks what adaptive_filter_for_rows predicted


@needs_cli
def test_web_frontend_contract_stdin_stdout(tmp_path):
    import pngloss_amd as P
    src = str(tmp_path / "in.png")
    _write_png(src, P.synth_rgba(120, 50, 5, 2))
    data = open(src, "rb").read()
    res = {}
    for tag, exe in (("ref", REF_CLI), ("hip", HIP_CLI)):
        r = subprocess.run([exe, "-s20", "-b2", "--strip", "-"], input=data, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-500:]
        res[tag] = r.stdout
    assert res["hip"] == res["ref"] and res["hip"][:8] == b"\x89PNG\r\n\x1a\n"


# ---- our own command line tool (pngloss_amd/cli): the batch driver -------------------------------------------

OUR_CLI = os.path.join(U.ROOT, "pngloss_amd", "cli", "pngloss")
needs_our_cli = pytest.mark.skipif(not (os.path.exists(OUR_CLI) and os.path.exists(REF_CLI)),
                                   reason="pngloss_amd/cli/pngloss or the reference CLI build is missing")


@needs_our_cli
def test_batch_cli_writes_what_the_reference_cli_writes_file_by_file(tmp_path):
    """One invocation, many files of all classes (one GPU batch) == the reference tool run once per file."""
    import pngloss_amd as P
    g = U.load_npz("suite_small.npz")
    inputs = {"david": g["david/in"], "rose": g["rose/in"], "tux": g["tux/in"]}
    for i, (w, h, m) in enumerate([(64, 48, 0), (130, 9, 2), (96, 64, 3), (33, 77, 4), (120, 50, 5), (1, 1, 1), (257, 3, 0)]):
        inputs[f"synth{i}"] = P.synth_rgba(w, h, m, i)
    ours_dir, ref_dir = tmp_path / "ours", tmp_path / "ref"
    ours_dir.mkdir(); ref_dir.mkdir()
    for name, arr in inputs.items():
        _write_png(str(ours_dir / f"{name}.png"), arr)
        _write_png(str(ref_dir / f"{name}.png"), arr)
    names = sorted(inputs)
    r = subprocess.run([OUR_CLI, "-v", "-s", "25", "-b", "3"] + [str(ours_dir / f"{n}.png") for n in names], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    assert f"Compressed {len(names)} images." in r.stderr
    for n in names:
        rr = subprocess.run([REF_CLI, "-s", "25", "-b", "3", str(ref_dir / f"{n}.png")], capture_output=True, timeout=300)
        assert rr.returncode == 0
        assert (ours_dir / f"{n}-loss.png").read_bytes() == (ref_dir / f"{n}-loss.png").read_bytes(), n
        assert f"{n}.png:" in r.stderr and "used " in r.stderr


@needs_our_cli
def test_batch_cli_pipe_ext_and_skip_if_larger(tmp_path):
    import pngloss_amd as P
    src = tmp_path / "in.png"
    _write_png(str(src), P.synth_rgba(90, 40, 5, 7))
    data = src.read_bytes()
    ours = subprocess.run([OUR_CLI, "-s20", "-b2", "--strip", "-"], input=data, capture_output=True, timeout=300)
    ref = subprocess.run([REF_CLI, "-s20", "-b2", "--strip", "-"], input=data, capture_output=True, timeout=300)
    assert ours.returncode == ref.returncode == 0 and ours.stdout == ref.stdout
    # --ext and per-file exit status: the second file does not exist -> code 2, the first is still written
    r = subprocess.run([OUR_CLI, "--ext", ".small.png", str(src), str(tmp_path / "missing.png")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and (tmp_path / "in.small.png").exists()
    # a tiny noise image grows when re-encoded at strength 0: --skip-if-larger must refuse it like the reference (98)
    noise = tmp_path / "noise.png"
    _write_png(str(noise), P.synth_rgba(8, 8, 1, 3))
    a = subprocess.run([OUR_CLI, "-f", "-s", "0", "--skip-if-larger", "-o", str(tmp_path / "o1.png"), str(noise)], capture_output=True, timeout=300)
    b = subprocess.run([REF_CLI, "-f", "-s", "0", "--skip-if-larger", "-o", str(tmp_path / "o2.png"), str(noise)], capture_output=True, timeout=300)
    assert a.returncode == b.returncode
    assert (tmp_path / "o1.png").exists() == (tmp_path / "o2.png").exists()


@needs_our_cli
def test_batch_cli_gpu_deflate_same_pixels_same_filters_not_larger(tmp_path):
    """--gpu-deflate: the IDAT bytes are not zlib's, everything a decoder sees is what the reference tool wrote:
    pixels, colour type, per-row filter bytes, ancillary chunks; and the files are not larger."""
    import io
    from PIL import Image
    import pngloss_amd as P
    g = U.load_npz("suite_small.npz")
    inputs = {"david": g["david/in"], "rose": g["rose/in"], "tux": g["tux/in"]}
    for i, (w, h, m) in enumerate([(640, 360, 0), (130, 9, 2), (96, 64, 3), (33, 77, 4), (320, 200, 5), (1, 1, 1), (700, 300, 1)]):
        inputs[f"synth{i}"] = P.synth_rgba(w, h, m, i)
    a_dir, b_dir = tmp_path / "zlib", tmp_path / "gpu"
    a_dir.mkdir(); b_dir.mkdir()
    for name, arr in inputs.items():
        _write_png(str(a_dir / f"{name}.png"), arr)
        _write_png(str(b_dir / f"{name}.png"), arr)
    names = sorted(inputs)
    ra = subprocess.run([OUR_CLI, "-s", "19", "-b", "2"] + [str(a_dir / f"{n}.png") for n in names], capture_output=True, text=True, timeout=600)
    rb = subprocess.run([OUR_CLI, "--gpu-deflate", "-v", "-s", "19", "-b", "2"] + [str(b_dir / f"{n}.png") for n in names], capture_output=True, text=True, timeout=600)
    assert ra.returncode == 0 and rb.returncode == 0, (ra.stderr[-400:], rb.stderr[-400:])
    total_a = total_b = 0
    for n in names:
        a = (a_dir / f"{n}-loss.png").read_bytes()
        b = (b_dir / f"{n}-loss.png").read_bytes()
        h, w = inputs[n].shape[:2]
        assert np.array_equal(np.array(Image.open(io.BytesIO(a)).convert("RGBA")), np.array(Image.open(io.BytesIO(b)).convert("RGBA"))), n
        assert _idat_filter_bytes(a, w, h) == _idat_filter_bytes(b, w, h), n
        assert a[:33] == b[:33]                                       # signature + IHDR
        total_a += len(a); total_b += len(b)
    assert total_b <= 1.005 * total_a
    # the pipe contract holds with the option as well
    data = (a_dir / "synth4.png").read_bytes()
    p1 = subprocess.run([OUR_CLI, "-s20", "-b2", "--strip", "-"], input=data, capture_output=True, timeout=300)
    p2 = subprocess.run([OUR_CLI, "--gpu-deflate", "-s20", "-b2", "--strip", "-"], input=data, capture_output=True, timeout=300)
    assert p1.returncode == p2.returncode == 0
    assert np.array_equal(np.array(Image.open(io.BytesIO(p1.stdout)).convert("RGBA")), np.array(Image.open(io.BytesIO(p2.stdout)).convert("RGBA")))


@needs_our_cli
def test_batch_cli_windows_are_pipelined_without_changing_the_outputs(tmp_path):
    """Many files = several GPU batches (windows); the next window is decoded while the current one is on the GPU.
    A window of 3 files (test hook PNGLOSS_WINDOW_FILES) must write exactly what one big window writes, with the
    per-file messages still in command line order."""
    import pngloss_amd as P
    a_dir, b_dir = tmp_path / "one", tmp_path / "many"
    a_dir.mkdir(); b_dir.mkdir()
    names = []
    for i, (w, h, m) in enumerate([(64, 48, 0), (130, 9, 2), (96, 64, 3), (33, 77, 4), (120, 50, 5), (80, 80, 1), (257, 3, 0), (50, 50, 2)]):
        arr = P.synth_rgba(w, h, m, i)
        for d in (a_dir, b_dir):
            _write_png(str(d / f"f{i}.png"), arr)
        names.append(f"f{i}")
    bad = "missing.png"                                   # a failing file in the middle keeps its place and exit code
    order = names[:4] + [None] + names[4:]
    runs = {}
    for d, env in ((a_dir, {}), (b_dir, {"PNGLOSS_WINDOW_FILES": "3"})):
        args = [str(d / (f"{n}.png" if n else bad)) for n in order]
        runs[d] = subprocess.run([OUR_CLI, "-v", "--gpu-deflate"] + args, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert runs[d].returncode == 2, runs[d].stderr[-500:]
    for n in names:
        assert (a_dir / f"{n}-loss.png").read_bytes() == (b_dir / f"{n}-loss.png").read_bytes(), n
    for r in runs.values():
        pos = [r.stderr.index(f"{n}.png:") for n in names[:4]] + [r.stderr.index("missing.png")] + [r.stderr.index(f"{n}.png:") for n in names[4:]]
        assert pos == sorted(pos)


@needs_our_cli
def test_batch_cli_gpu_read_writes_the_same_files(tmp_path):
    """--gpu-read: chunk walk + inflate on the decode threads, inverse filters + expansion to RGBA8 on the device.  Files of
    every colour type / bit depth / tRNS (fixtures written by tests/golden/make_png_read_golden.py), three suite files and one file with
    a text chunk (which must fall back to libpng): the outputs must be byte for byte what the tool writes without the option -- and
    therefore what the reference tool writes."""
    from PIL import Image, PngImagePlugin
    import pngloss_amd as P
    fx = U.png_read_fixtures()
    pick = [f for f in fx if f[0].startswith("suite_rose") or f[0].startswith("suite_tux") or f[0].startswith("suite_david") or "37x19" in f[0] or "130x70" in f[0]][:40]
    a_dir, b_dir = tmp_path / "libpng", tmp_path / "gpu"
    a_dir.mkdir(); b_dir.mkdir()
    names = []
    for name, png, _ in pick:
        for d in (a_dir, b_dir):
            (d / f"{name}.png").write_bytes(png)
        names.append(name)
    inter = P.synth_rgba(70, 33, 5, 1)
    meta = PngImagePlugin.PngInfo(); meta.add_text("Comment", "made for the fallback test")
    for d in (a_dir, b_dir):
        Image.fromarray(inter, "RGBA").save(str(d / "withtext.png"), pnginfo=meta)
    names.append("withtext")
    ra = subprocess.run([OUR_CLI, "-s", "19", "-b", "2"] + [str(a_dir / f"{n}.png") for n in names], capture_output=True, text=True, timeout=600)
    rb = subprocess.run([OUR_CLI, "--gpu-read", "-s", "19", "-b", "2"] + [str(b_dir / f"{n}.png") for n in names], capture_output=True, text=True, timeout=600)
    assert ra.returncode == 0 and rb.returncode == 0, (ra.stderr[-600:], rb.stderr[-600:])
    for n in names:
        assert (a_dir / f"{n}-loss.png").read_bytes() == (b_dir / f"{n}-loss.png").read_bytes(), n
