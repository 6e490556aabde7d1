#!/bin/bash
# Developer tool (GPU box): bench.py twice (optionally with environment for the second run), the legs that matter in one line each
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05bl}
cd /tmp && export TMPDIR=/tmp
cd $R
nproc > $OUT/${TAG}_legs.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/${TAG}_legs.txt 2>/dev/null
i=0
for ENVS in "${@:2}"; do
  i=$((i+1))
  env $(echo "$ENVS" | tr ',' ' ') python bench.py ${BENCH_FLAGS:-} > $OUT/${TAG}_bench$i.json 2> $OUT/${TAG}_bench$i.err
  python - $OUT/${TAG}_bench$i.json "$ENVS" >> $OUT/${TAG}_legs.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("===", sys.argv[2])
print("headline", d["value"], "suite", d.get("suite_batch", {}).get("value"), "batch256", d.get("batch", {}).get("value"), "sat", d.get("batch_saturating", {}).get("value"))
print("shares", [(s["frames"], s["value"], s["engine"][:3]) for s in d.get("batch_rank_share", {}).get("shares", [])], d.get("batch_rank_share", {}).get("projected_strong_scaling"))
print("sweep", [(p["strength"], p["bleed"], p["value"]) for p in d.get("sweep_8192", {}).get("points", [])])
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("full_frame"))
PY
done
