#!/bin/bash
# pl_hist: kernel time on the 4096x4096 frame and on 64 x 1080p (rocprofv3 kernel trace); PNGLOSS_HIP_HIST=1 = round 2's shape (8 replicas,
# 256 threads).  The other variants of profiles/r03_hist_variants.txt were template instances of launch_hist_v (pl_prepost.hip) behind the same switch.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
python -m pytest tests/test_gpu_parity.py -q -k "histogram" 2>&1 | tail -3
for v in ${HIST_VARIANTS:-0 1}; do
  rm -rf /tmp/hv$v
  PNGLOSS_HIP_HIST=$v rocprofv3 --kernel-trace --stats -d /tmp/hv$v -o t --output-format csv -- python tests/tools/gpu_seg_time.py 4096 4096 0 19 2 2 > /tmp/hv$v.log 2>&1
  echo "variant $v: $(grep -h pl_hist $(find /tmp/hv$v -name '*kernel_stats.csv') | cut -c1-160)  | $(tail -1 /tmp/hv$v.log | cut -c30-200)"
done
# batch: 64 x 1080p frames, workgroup engine
cat > /tmp/hb.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import pngloss_amd as P
w, h, n = 1920, 1080, 64
ctx = P.HipContext()
base = [P.synth_rgba(w, h, 0, i) for i in range(4)]
for rep in range(2):
    ds = [torch.from_numpy(base[i % 4].copy()).cuda() for i in range(n)]
    fs = [torch.zeros(h, dtype=torch.uint8, device="cuda") for i in range(n)]
    torch.cuda.synchronize()
    ctx.run([(d.data_ptr(), f.data_ptr(), w, h) for d, f in zip(ds, fs)], 19, 2)
    print("n=%d %.2f ms out0=%016x" % (n, ctx.engine_ms, P.fnv1a64(ds[0].cpu().numpy(), P.SURVEY_FNV_BASIS)))
PY
for v in 0 1; do
  rm -rf /tmp/hb$v
  PNGLOSS_HIP_HIST=$v PNGLOSS_HIP_ENGINE=wg rocprofv3 --kernel-trace --stats -d /tmp/hb$v -o t --output-format csv -- python /tmp/hb.py > /tmp/hb$v.log 2>&1
  echo "batch variant $v: $(grep -h pl_hist $(find /tmp/hb$v -name '*kernel_stats.csv') | cut -c1-160) | $(tail -1 /tmp/hb$v.log | cut -c1-160)"
done
