#!/bin/bash
# Developer tool (GPU box): A/B timing of two builds of the row engine inside ONE gpurun call (boxes and consecutive runs differ
# by several percent, so only interleaved runs compare).  usage: tools/ab_test.sh "<flags A>" "<flags B>" [rounds]
# env AB_W / AB_H: frame size (default 4096x1024, the top of the headline frame).  Prints engine ms per run, then min/median.
cd ${GRAFT_REPO_ROOT:-.}
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
R=${3:-8}
for v in A B; do
  if [ $v = A ]; then F="$1"; else F="$2"; fi
  touch pngloss_amd/csrc/pl_engine.hip
  make -C pngloss_amd/csrc HIPFLAGS="$BASE $F" > /dev/null 2>&1
  mkdir -p /tmp/ab_$v; cp pngloss_amd/csrc/libpngloss_hip.so /tmp/ab_$v/
done
rm -f /tmp/ab_A.txt /tmp/ab_B.txt
for i in $(seq 1 $R); do
  for v in A B; do
    cp /tmp/ab_$v/libpngloss_hip.so pngloss_amd/csrc/libpngloss_hip.so
    PNGLOSS_HIP_DEBUG=1 python tools/lead_time.py ${AB_W:-4096} ${AB_H:-1024} 2>&1 | grep "engine [0-9.]* ms" | tail -1 | sed 's/.*engine \([0-9.]*\) ms.*/\1/' >> /tmp/ab_$v.txt
  done
done
for v in A B; do
  echo -n "$v: "; sort -n /tmp/ab_$v.txt | tr '\n' ' '; echo
  sort -n /tmp/ab_$v.txt | awk '{a[NR]=$1} END {printf "   min %.1f  median %.1f  n=%d\n", a[1], (NR%2? a[(NR+1)/2] : (a[NR/2]+a[NR/2+1])/2), NR}'
done
