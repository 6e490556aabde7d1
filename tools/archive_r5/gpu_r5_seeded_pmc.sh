#!/bin/bash
# Developer tool (GPU box): SQ counters of the segment engine's kernels on a seeded strip (8192 x 256, s=85 b=2), two passes of eight counters; sums per kernel name
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT/r05p_seeded_pmc.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
n=0
for PMC in "$P1" "$P2"; do
  n=$((n+1)); rm -rf /tmp/pmc$n
  ( cd $R && rocprofv3 --pmc $PMC -d /tmp/pmc$n -o p --output-format csv -- python tests/tools/gpu_seg_time.py 8192 256 0 ${SB:-85 2} 1 > /tmp/run$n.log 2>&1 )
  grep engine /tmp/run$n.log | tail -1 >> $OUT/r05p_seeded_pmc.txt
  F=$(find /tmp/pmc$n -name "*counter_collection.csv" | head -1)
  python - "$F" >> $OUT/r05p_seeded_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(k, r["Counter_Name"])] += 1
for k in acc:
    if "seg_k" not in k: continue
    n = max(calls[(k, c)] for c in acc[k])
    print("%-42s launches %5d  " % (k, n) + "  ".join("%s %.0f" % (c, v / n) for c, v in sorted(acc[k].items())))
PY
done
