"""CPU tests of the C host side of the command line tool (no GPU compute):
  * our rwpng.c (PNG <-> RGBA8, explicit row filters) against the reference's rwpng.c on PNGs of every colour type --
    both compiled into the same tiny driver (tests/c/rwpng_copy.c), outputs must be byte-identical;
  * the option handling / exit codes of pngloss_amd/cli/pngloss against the reference CLI for argument errors."""
import io
import os
import subprocess

import numpy as np
import pytest

from tests import util as U

CLI = os.path.join(U.ROOT, "pngloss_amd", "cli")
PNG_INC = "/opt/conda/include"
PNG_LIB = "/lib/x86_64-linux-gnu/libpng16.so.16"
REF_SRC = "/root/reference/src"
REF_CLI = os.path.join(U.ROOT, "oracle", "_ref", "pngloss_ref_cli")
have_png = os.path.exists(os.path.join(PNG_INC, "png.h")) and os.path.exists(PNG_LIB)
pytestmark = pytest.mark.skipif(not have_png, reason="libpng headers/runtime not found on this box")


@pytest.fixture(scope="module")
def drivers(tmp_path_factory):
    d = tmp_path_factory.mktemp("rwpng")
    ours = str(d / "copy_ours")
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-w", "-I" + PNG_INC, "-I" + CLI, "-o", ours,
                    os.path.join(U.ROOT, "tests", "c", "rwpng_copy.c"), os.path.join(CLI, "png_bridge.c"), PNG_LIB, "-lz", "-lm"], check=True)
    ref = None
    if os.path.exists(os.path.join(REF_SRC, "rwpng.c")):
        ref = str(d / "copy_ref")
        subprocess.run(["gcc", "-O1", "-std=gnu11", "-w", '-DRWPNG_HEADER="rwpng.h"', "-I" + PNG_INC, "-I" + REF_SRC, "-o", ref,
                        os.path.join(U.ROOT, "tests", "c", "rwpng_copy.c"), os.path.join(REF_SRC, "rwpng.c"), PNG_LIB, "-lz", "-lm"], check=True)
    return ours, ref, d


def _samples(d):
    from PIL import Image, PngImagePlugin
    rng = np.random.default_rng(4)
    out = []
    rgba = rng.integers(0, 256, (23, 31, 4), dtype=np.uint8)

    def save(name, im, **kw):
        p = str(d / name)
        im.save(p, **kw)
        out.append(p)

    save("rgba.png", Image.fromarray(rgba, "RGBA"))
    save("rgb.png", Image.fromarray(rgba[..., :3].copy(), "RGB"))
    save("gray.png", Image.fromarray(rgba[..., 0].copy(), "L"))
    save("graya.png", Image.fromarray(rgba[..., :2].copy(), "LA"))
    pal = Image.fromarray(rgba[..., :3].copy(), "RGB").quantize(16)
    save("palette.png", pal)
    save("palette_trns.png", pal, transparency=3)
    save("gray16.png", Image.fromarray((rng.integers(0, 65536, (9, 14))).astype(np.uint16)))
    save("bilevel.png", Image.fromarray(rgba[..., 0] > 128))
    save("rgb_interlaced_gamma.png", Image.fromarray(rgba[..., :3].copy(), "RGB"), gamma=0.5)
    meta = PngImagePlugin.PngInfo()
    meta.add_text("Comment", "carried through unless --strip")
    save("rgba_text_dpi.png", Image.fromarray(rgba, "RGBA"), pnginfo=meta, dpi=(72, 72))
    opaque_gray = np.stack([rgba[..., 1]] * 3 + [np.full((23, 31), 255, np.uint8)], axis=-1)
    save("rgba_that_is_gray.png", Image.fromarray(opaque_gray, "RGBA"))
    return out


def test_our_rwpng_decodes_like_pil_and_reencodes_losslessly(drivers):
    from PIL import Image
    ours, _, d = drivers
    for src in _samples(d):
        for policy in (-1, 1, 5):
            dst = src[:-4] + f".ours{policy}.png"
            r = subprocess.run([ours, src, dst, str(policy), "0"], capture_output=True, text=True)
            assert r.returncode == 0, (src, r.stderr)
            want = np.array(Image.open(src).convert("RGBA"))
            if "gray16" in src:
                want = None          # PIL's 16->8 conversion differs from libpng's strip_16; covered by the ref comparison
            got = np.array(Image.open(dst).convert("RGBA"))
            if want is not None:
                assert np.array_equal(got, want), (src, policy)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_SRC, "rwpng.c")), reason="needs /root/reference (build container)")
def test_our_rwpng_writes_the_same_bytes_as_the_reference_rwpng(drivers):
    ours, ref, d = drivers
    for src in _samples(d):
        for policy in (-1, 0, 1, 2, 3, 4, 5):
            for strip in (0, 1):
                a, b = src[:-4] + ".a.png", src[:-4] + ".b.png"
                ra = subprocess.run([ours, src, a, str(policy), str(strip)], capture_output=True, text=True)
                rb = subprocess.run([ref, src, b, str(policy), str(strip)], capture_output=True, text=True)
                assert ra.returncode == rb.returncode == 0, (src, ra.stderr, rb.stderr)
                assert ra.stdout == rb.stdout, (src, policy, strip, ra.stdout, rb.stdout)
                assert open(a, "rb").read() == open(b, "rb").read(), (src, policy, strip)


def _build_cli():
    exe = os.path.join(CLI, "pngloss")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", CLI], check=True, capture_output=True)
    return exe


@pytest.mark.skipif(not os.path.exists(REF_CLI), reason="oracle/_ref/pngloss_ref_cli not built")
@pytest.mark.parametrize("args", [[], ["-s", "300", "x.png"], ["-b", "0", "x.png"], ["-b", "40000", "x.png"], ["-s", "abc", "x.png"],
                                  ["-o", "a.png", "-o", "b.png", "x.png"], ["--ext", "e.png", "-o", "a.png", "x.png"],
                                  ["-o", "a.png", "x.png", "y.png"], ["-o", "-", "x.png", "y.png"], ["--bogus"],
                                  ["-f", "/nonexistent/dir/file.png"], ["-V"]])
def test_argument_errors_and_exit_codes_match_the_reference(args, tmp_path):
    exe = _build_cli()
    ours = subprocess.run([exe] + args, capture_output=True, text=True, cwd=tmp_path)
    ref = subprocess.run([REF_CLI] + args, capture_output=True, text=True, cwd=tmp_path)
    assert ours.returncode == ref.returncode, (args, ours.stderr, ref.stderr)
    if args and args != ["-V"] and args != ["--bogus"]:
        assert ours.stderr.strip().splitlines()[:1] == ref.stderr.strip().splitlines()[:1]


def test_overwrite_rule_is_checked_before_any_work(tmp_path):
    exe = _build_cli()
    from PIL import Image
    src = tmp_path / "a.png"
    Image.fromarray(np.zeros((4, 4, 4), np.uint8), "RGBA").save(src)
    (tmp_path / "a-loss.png").write_bytes(b"existing")
    r = subprocess.run([exe, str(src)], capture_output=True, text=True)
    assert r.returncode == 15 and "exists; not overwriting" in r.stderr
    assert (tmp_path / "a-loss.png").read_bytes() == b"existing"


def test_stream_writer_is_byte_identical_to_libpng(drivers):
    """png_stream_writer.c (PNG container + zlib around pre-filtered scanlines, no libpng) against libpng's writer on
    every colour type, with fixed, cycling and heuristic filters, with and without ancillary chunks, and on sizes that
    exercise the zlib window / CMF rules for tiny images and the 8192-byte IDAT slicing for bigger ones."""
    from PIL import Image
    _, _, d = drivers
    exe = str(d / "stream_copy")
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-w", "-I" + PNG_INC, "-I" + CLI, "-o", exe, os.path.join(U.ROOT, "tests", "c", "stream_copy.c"),
                    os.path.join(CLI, "png_bridge.c"), os.path.join(CLI, "png_stream_writer.c"), PNG_LIB, "-lz", "-lm"], check=True)
    files = _samples(d)
    rng = np.random.default_rng(8)
    # (1-pixel-wide and 1-pixel-high images: libpng refuses the filters that have no neighbour there)
    for k, (w, h) in enumerate([(1, 1), (3, 2), (40, 30), (90, 61), (300, 200), (1, 20), (1, 2), (1, 300), (20, 1), (300, 1)]):      # 4 B .. 240 KB of scanlines
        p = str(d / f"size{k}.png")
        Image.fromarray((rng.integers(0, 256, (h, w, 4)) // 32 * 32).astype(np.uint8), "RGBA").save(p)
        files.append(p)
    for src in files:
        for policy in (-1, 0, 1, 3, 4, 5):
            for strip in (0, 1):
                a, b = src[:-4] + ".s.png", src[:-4] + ".l.png"
                r = subprocess.run([exe, src, a, b, str(policy), str(strip)], capture_output=True, text=True)
                assert r.returncode == 0, (src, policy, r.stdout, r.stderr)
                n1, n2, m1, m2 = map(int, r.stdout.split())
                assert open(a, "rb").read() == open(b, "rb").read(), (src, policy, strip)
                assert n1 == n2 and m1 == m2
