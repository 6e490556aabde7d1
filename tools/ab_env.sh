#!/bin/bash
# Developer tool (GPU box): like ab_so.sh, but the variants are environment settings of one build: tools/ab_env.sh "VAR=1" "VAR=2" ...
cd ${GRAFT_REPO_ROOT:-.}
R=${AB_ROUNDS:-10}
n=0; for F in "$@"; do n=$((n+1)); rm -f /tmp/abenv_$n.txt; done
for i in $(seq 1 $R); do
  v=0
  for F in "$@"; do v=$((v+1))
    env $F PNGLOSS_HIP_DEBUG=1 python tools/lead_time.py ${AB_W:-4096} ${AB_H:-1024} ${AB_MODE:-0} 2>&1 | grep "engine [0-9.]* ms" | tail -1 | sed 's/.*engine \([0-9.]*\) ms.*/\1/' >> /tmp/abenv_$v.txt
  done
done
v=0; for F in "$@"; do v=$((v+1)); sort -n /tmp/abenv_$v.txt | awk -v f="$F" '{a[NR]=$1} END {printf "min %.1f  median %.1f  n=%d   [%s]\n", a[1], (NR%2? a[(NR+1)/2] : (a[NR/2]+a[NR/2+1])/2), NR, f}'; done
