#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
make -C pngloss_amd/cli >/dev/null 2>&1
D=/tmp/clis; rm -rf $D; mkdir -p $D/a $D/b
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import pngloss_amd as P
from PIL import Image
Image.fromarray(P.synth_rgba(4096, 4096, 0, 0), "RGBA").save("/tmp/clis/a/big.png", compress_level=1)
for i in range(32): Image.fromarray(P.synth_rgba(1280, 720, 0, i), "RGBA").save("/tmp/clis/b/f%02d.png" % i, compress_level=1)
PY
cd /tmp; export TMPDIR=/tmp
for w in a b; do
rm -rf /tmp/rp_$w
PNGLOSS_TIMING=1 rocprofv3 --kernel-trace --stats -d /tmp/rp_$w -o t --output-format csv -- $R/pngloss_amd/cli/pngloss -f --gpu-read --gpu-deflate --ext -gpu.png $D/$w/*.png 2>&1 | grep timing
grep -h "pr_k\|pl_hist\|Name" $(find /tmp/rp_$w -name "*kernel_stats.csv") | cut -c1-150
done
