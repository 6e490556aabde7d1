#!/bin/bash
# Developer tool (GPU box): TIMING-ONLY builds of the seeded enumeration with one phase left out each (results are wrong; only the kernel's own duration is read):
# rocprofv3 --kernel-trace of an 8192 x 64 strip at s=85 b=2 and s=40 b=1 per variant (libraries tools/ablate_build/libpngloss_hip_abl*.so, built from a temporary patch)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $R
cp pngloss_amd/csrc/libpngloss_hip.so /tmp/keep.so
: > $OUT/r05a_seeded_ablate.txt
for V in base ablseg ablb abla ablx ablsb ablall; do
  [ $V = base ] && cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so || cp tools/ablate_build/libpngloss_hip_$V.so pngloss_amd/csrc/libpngloss_hip.so
  for SB in "85 2" "40 1"; do
    rm -rf /tmp/prof
    timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t --output-format csv -- python tests/tools/gpu_seg_time.py 8192 64 0 $SB 1 > /tmp/run.log 2>&1
    F=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
    echo "=== $V  s b = $SB   $(grep engine /tmp/run.log | tail -1 | cut -c1-70)" >> $OUT/r05a_seeded_ablate.txt
    [ -n "$F" ] && python - "$F" >> $OUT/r05a_seeded_ablate.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "seg_k" in r["Name"]: print("   %-50s calls %6s avg %9.2f us" % (r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so
