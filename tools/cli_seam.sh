#!/bin/bash
# Where a command-line run spends its time (PNGLOSS_TIMING / PNGLOSS_HIP_DEBUG_SEAM prints): one large file, the same again, a window of 32 files.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
make -C pngloss_amd/cli >/dev/null 2>&1
D=/tmp/clis; rm -rf $D; mkdir -p $D
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import pngloss_amd as P
from PIL import Image
Image.fromarray(P.synth_rgba(4096, 4096, 0, 0), "RGBA").save("/tmp/clis/big.png", compress_level=1)
for i in range(32): Image.fromarray(P.synth_rgba(1280, 720, 0, i), "RGBA").save("/tmp/clis/f%02d.png" % i, compress_level=1)
PY
export PNGLOSS_TIMING=1 PNGLOSS_HIP_DEBUG_SEAM=1 PNGLOSS_HIP_DEBUG=1
echo "== one 4096x4096 file, --gpu-deflate"; ( time pngloss_amd/cli/pngloss -f --gpu-deflate --ext -gpu.png $D/big.png ) 2>&1 | grep -v "^$" | cut -c1-260
rm -f $D/*-gpu.png; echo "== the same again"; ( time pngloss_amd/cli/pngloss -f --gpu-deflate --ext -gpu.png $D/big.png ) 2>&1 | grep -v "^$" | cut -c1-260
rm -f $D/*-gpu.png; echo "== 32 files 1280x720, --gpu-deflate"; ( time pngloss_amd/cli/pngloss -f --gpu-deflate --ext -gpu.png $D/f*.png ) 2>&1 | grep -v "^$" | cut -c1-260
rm -f $D/*-gpu.png; echo "== one 4096x4096 file, --gpu-read --gpu-deflate"; ( time pngloss_amd/cli/pngloss -f --gpu-read --gpu-deflate --ext -gpu.png $D/big.png ) 2>&1 | grep -v "^$" | cut -c1-260
rm -f $D/*-gpu.png; echo "== 32 files 1280x720, --gpu-read --gpu-deflate"; ( time pngloss_amd/cli/pngloss -f --gpu-read --gpu-deflate --ext -gpu.png $D/f*.png ) 2>&1 | grep -v "^$" | cut -c1-260
