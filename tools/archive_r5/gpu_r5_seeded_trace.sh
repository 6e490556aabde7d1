#!/bin/bash
# Developer tool (GPU box): rocprofv3 --kernel-trace --stats of the seeded points named on the command line ("S B" pairs) on an 8192 x 1024 strip
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT/r05y_seeded_trace.txt
for SB in "$@"; do
  rm -rf /tmp/prof; 
  ( cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof -o t --output-format csv -- python tests/tools/gpu_seg_time.py 8192 1024 0 $SB 2 > /tmp/run.log 2>&1 )
  echo "=== s b = $SB" >> $OUT/r05y_seeded_trace.txt
  grep engine /tmp/run.log | tail -1 >> $OUT/r05y_seeded_trace.txt
  F=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
  python - "$F" >> $OUT/r05y_seeded_trace.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-60s calls %6s avg %9.2f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
