#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of prebuilt libraries on one frame, same box: tools/gpu_r5_ab_kern.sh tag lib...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $R
cp pngloss_amd/csrc/libpngloss_hip.so /tmp/keep.so
: > $OUT/${TAG}_abkern.txt
for rep in 1; do
for L in base "$@"; do
  [ $L = base ] && cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so || cp tools/ablate_build/libpngloss_hip_$L.so pngloss_amd/csrc/libpngloss_hip.so
  rm -rf /tmp/abk; rocprofv3 --kernel-trace --stats -d /tmp/abk -o t --output-format csv -- python tests/tools/gpu_seg_time.py ${AB_W:-4096} ${AB_H:-2048} 0 19 2 2 > /dev/null 2>&1
  echo "=== $L (rep $rep)" >> $OUT/${TAG}_abkern.txt
  python - "$(find /tmp/abk -name "*kernel_stats.csv" | head -1)" >> $OUT/${TAG}_abkern.txt <<'PY'
import csv, sys
tot = 0.0
for r in csv.reader(open(sys.argv[1])):
    if len(r) > 4 and "seg_k_" in r[0] and "resolve" not in r[0]:
        n = r[0].split("seg_k_")[1].split("(")[0]
        print("seg_k_%-24s calls %6d avg %8.1f ns" % (n, int(r[1]), float(r[3]))); tot += float(r[3])
print("sum of the four averages %.1f ns" % tot)
PY
done; done
cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so
