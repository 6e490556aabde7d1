#!/bin/bash
# round 6: where a command-line run spends its time (PNGLOSS_TIMING=1 PNGLOSS_HIP_DEBUG_SEAM=1): one 4096x4096 RGBA file, a window of 32 files 1280x720, the reference's
# eleven suite files -- the default path (libpng / zlib on host threads), --gpu-deflate, --gpu-read --gpu-deflate; best of the runs shown (every case twice)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out
make -C pngloss_amd/cli >/dev/null 2>&1
D=/tmp/clis; rm -rf $D; mkdir -p $D/suite
python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import pngloss_amd as P
from PIL import Image
Image.fromarray(P.synth_rgba(4096, 4096, 0, 0), "RGBA").save("/tmp/clis/big.png", compress_level=1)
for i in range(32): Image.fromarray(P.synth_rgba(1280, 720, 0, i), "RGBA").save("/tmp/clis/f%02d.png" % i, compress_level=1)
z = np.load("tests/golden/suite_png.npz")
for k in z.files: open("/tmp/clis/suite/%s.png" % k, "wb").write(z[k].tobytes())
PY
export PNGLOSS_TIMING=1 PNGLOSS_HIP_DEBUG_SEAM=1
run() { # label, files..., then flags after --
  local label=$1; shift
  for rep in 1 2; do
    rm -f $D/*-out.png $D/suite/*-out.png
    echo "== $label (run $rep): pngloss -f $FLAGS"; ( time pngloss_amd/cli/pngloss -f $FLAGS --ext -out.png "$@" ) 2>&1 | grep -v "^$" | grep "timing\|real\|deflate stage\|read side\|host window chunk\|user\|sys" | cut -c1-230
  done
}
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}; bash tools/gpu_r6_cli.sh (one MI355X, gpurun box; host: $(nproc) logical CPUs visible)"
  for FLAGS in "" "--gpu-deflate" "--gpu-read --gpu-deflate"; do
    run "one 4096x4096 file" $D/big.png
    run "32 files 1280x720" $D/f*.png
    run "the reference's 11 suite files" $D/suite/*.png
  done
} > $OUT/r06_cli.txt 2>&1
