#!/bin/bash
# round 6: the device inflate (pl_inflate_core.h: rounds of 64 speculative table look-ups, the chain followed in scalar registers): parity on the device, MB/s per stream, aggregate
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  echo "## python -m pytest tests/test_png_read.py -m gpu -q"; timeout 900 python -m pytest tests/test_png_read.py -m gpu -q 2>&1 | tail -3
  for n in ${INFL_N:-1 32 256 512 768}; do
    echo "## READ_DISTINCT=32 python tests/tools/gpu_read_time.py $n 1280 720 16"; READ_DISTINCT=32 timeout 600 python tests/tools/gpu_read_time.py $n 1280 720 16 2>&1 | grep -v amdgpu.ids | tail -4
  done
  echo "## python tests/tools/gpu_inflate_suite.py"; timeout 600 python tests/tools/gpu_inflate_suite.py 2>&1 | grep -v amdgpu.ids | tail -14
  echo "## one 4096x4096 file"; timeout 600 python tests/tools/gpu_read_time.py 1 4096 4096 16 2>&1 | grep -v amdgpu.ids | tail -2
  echo "## rocprofv3 --kernel-trace --stats -- python tests/tools/gpu_read_time.py 32 1280 720 16"
  rm -rf $OUT/infl_prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/infl_prof -o infl -- python tests/tools/gpu_read_time.py 32 1280 720 16 > /dev/null 2>&1
  python - <<'P'
import csv, glob, os
for f in glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/infl_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "inflate" in r["Name"] or "pr_k" in r["Name"]: print(r["Name"][:60], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
P
} > $OUT/r06_inflate.txt 2>&1
