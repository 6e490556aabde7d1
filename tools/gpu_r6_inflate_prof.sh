#!/bin/bash
# round 6: rocprofv3 kernel statistics of the read side on 32 files of 1280x720 (pl_inflate_k, pr_k_decode)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  echo "## rocprofv3 --kernel-trace --stats --output-format csv -- python tests/tools/gpu_read_time.py 32 1280 720 16   (READ_DISTINCT=8)"
  rm -rf $OUT/infl_prof; READ_DISTINCT=8 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/infl_prof -o infl -- python tests/tools/gpu_read_time.py 32 1280 720 16 > /dev/null 2>&1
  python - <<'P'
import csv, glob, os
for f in glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/infl_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %5s  avg %12.1f us  min %12.1f  max %12.1f  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
P
} > $OUT/r06_inflate_prof.txt 2>&1
