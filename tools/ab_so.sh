#!/bin/bash
# Developer tool (GPU box): like ab_test.sh, but compares prebuilt libraries: tools/ab_so.sh lib1.so lib2.so ...
cd ${GRAFT_REPO_ROOT:-.}
R=${AB_ROUNDS:-10}
cp pngloss_amd/csrc/libpngloss_hip.so /tmp/ab_keep.so
n=0; for F in "$@"; do n=$((n+1)); cp $F /tmp/abso_$n.so; rm -f /tmp/abso_$n.txt; done
for i in $(seq 1 $R); do
  for v in $(seq 1 $n); do
    cp /tmp/abso_$v.so pngloss_amd/csrc/libpngloss_hip.so
    PNGLOSS_HIP_DEBUG=1 python tools/lead_time.py ${AB_W:-4096} ${AB_H:-1024} 2>&1 | grep "engine [0-9.]* ms" | tail -1 | sed 's/.*engine \([0-9.]*\) ms.*/\1/' >> /tmp/abso_$v.txt
  done
done
v=0; for F in "$@"; do v=$((v+1)); sort -n /tmp/abso_$v.txt | head -6 | tr "\n" " "; echo; sort -n /tmp/abso_$v.txt | awk -v f="$F" '{a[NR]=$1} END {printf "min %.1f  median %.1f  n=%d   [%s]\n", a[1], (NR%2? a[(NR+1)/2] : (a[NR/2]+a[NR/2+1])/2), NR, f}'; done
cp /tmp/ab_keep.so pngloss_amd/csrc/libpngloss_hip.so
