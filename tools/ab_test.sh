#!/bin/bash
# Developer tool (GPU box): timing of several builds of the row engine inside ONE gpurun call, runs interleaved.  The GPU's
# clock sits on one of a few discrete levels per run (3.4 % apart), so the MINIMUM over >= 8 short runs -- the top level -- is
# the number to compare (reproducible to ~0.2 % inside a call; boxes differ by ~1 %).
# usage: tools/ab_test.sh "<flags variant 1>" "<flags variant 2>" ...     env: AB_ROUNDS (10), AB_W x AB_H (4096x1024)
cd ${GRAFT_REPO_ROOT:-.}
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
R=${AB_ROUNDS:-10}
n=0
for F in "$@"; do
  n=$((n+1))
  touch pngloss_amd/csrc/pl_engine.hip
  make -C pngloss_amd/csrc HIPFLAGS="$BASE $F" > /dev/null 2>&1 || echo "build of variant $n failed"
  mkdir -p /tmp/ab_$n; cp pngloss_amd/csrc/libpngloss_hip.so /tmp/ab_$n/; rm -f /tmp/ab_$n.txt
done
for i in $(seq 1 $R); do
  for v in $(seq 1 $n); do
    cp /tmp/ab_$v/libpngloss_hip.so pngloss_amd/csrc/libpngloss_hip.so
    PNGLOSS_HIP_DEBUG=1 python tools/lead_time.py ${AB_W:-4096} ${AB_H:-1024} 2>&1 | grep "engine [0-9.]* ms" | tail -1 | sed 's/.*engine \([0-9.]*\) ms.*/\1/' >> /tmp/ab_$v.txt
  done
done
v=0
for F in "$@"; do
  v=$((v+1))
  sort -n /tmp/ab_$v.txt | awk -v f="$F" '{a[NR]=$1} END {printf "min %.1f  median %.1f  n=%d   [%s]\n", a[1], (NR%2? a[(NR+1)/2] : (a[NR/2]+a[NR/2+1])/2), NR, f}'
done
