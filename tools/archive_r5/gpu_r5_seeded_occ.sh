#!/bin/bash
# Developer tool (GPU box): the seeded enumeration at 4 / 3 / 2 / 1 workgroups a CU (a TIMING build that pads the kernel's LDS request by PNGLOSS_HIP_ENUM_PAD bytes; never shipped):
# is the kernel bound by its 2.5 rounds of workgroup slots?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $R
cp pngloss_amd/csrc/libpngloss_hip.so /tmp/keep.so
cp tools/ablate_build/libpngloss_hip_pad.so pngloss_amd/csrc/libpngloss_hip.so
: > $OUT/r05o_seeded_occ.txt
for PAD in 0 12000 20000 28000; do
  for SB in "85 2" "40 1"; do
    rm -rf /tmp/prof
    PNGLOSS_HIP_ENUM_PAD=$PAD timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t --output-format csv -- python tests/tools/gpu_seg_time.py 8192 256 0 $SB 1 > /tmp/run.log 2>&1
    F=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
    echo "=== pad $PAD  s b = $SB   $(grep engine /tmp/run.log | tail -1 | cut -c1-70)" >> $OUT/r05o_seeded_occ.txt
    [ -n "$F" ] && python - "$F" >> $OUT/r05o_seeded_occ.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "enum_seeded" in r["Name"]: print("   %-50s calls %6s avg %9.2f us" % (r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so
