#!/bin/bash
# Developer tool (GPU box): the workgroup-per-image engine, library variants side by side: single 1080p frames 0..7 (engine ms, repaired / light pixels), then batches; parity tests of that engine
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $R
cp pngloss_amd/csrc/libpngloss_hip.so /tmp/keep.so
: > $OUT/${TAG}_wgab.txt
cat > /tmp/w.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import pngloss_amd as P
w, h = 1920, 1080
ctx = P.HipContext()
for i in range(8):
    a = P.synth_rgba(w, h, 0, i)
    d = torch.from_numpy(a.copy()).cuda(); f = torch.zeros(h, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.run([(d.data_ptr(), f.data_ptr(), w, h)], 19, 2)
    print("frame %d engine ms %.1f  out=%016x filt=%016x" % (i, ctx.engine_ms, P.fnv1a64(d.cpu().numpy(), P.SURVEY_FNV_BASIS), P.fnv1a64(f.cpu().numpy(), P.SURVEY_FNV_BASIS)), flush=True)
PY
for V in "$@"; do
  if [ "$V" = base ]; then cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so; else cp tools/ablate_build/libpngloss_hip_$V.so pngloss_amd/csrc/libpngloss_hip.so; fi
  echo "=== $V" >> $OUT/${TAG}_wgab.txt
  PNGLOSS_HIP_ENGINE=wg PNGLOSS_HIP_DEBUG=1 python /tmp/w.py 2>&1 | grep -E "repaired pixels|light pixels|engine ms" | sed 's/pngloss_hip: image 0: chain kcycles per wave/  kcycles/; s/pngloss_hip:   light pixels per chain wave/  light/; s/; rows on.*//' | cut -c1-200 >> $OUT/${TAG}_wgab.txt
  SEG_BATCH_ENGINES=wg python tests/tools/gpu_seg_batch.py 1920 1080 32 128 256 2>&1 | grep "n=" >> $OUT/${TAG}_wgab.txt
done
cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so
( PNGLOSS_HIP_ENGINE=wg timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3 ) >> $OUT/${TAG}_wgab.txt
