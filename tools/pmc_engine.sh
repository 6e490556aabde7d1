#!/bin/bash
# GPU box: SQ counters of pl_engine on one 1920x1080 (or $1 x $2) frame, one rocprofv3 --pmc pass per counter group.
# usage: tools/pmc_engine.sh [W H] -> gpurun_out/pmc_engine.txt
W=${1:-1920}; H=${2:-1080}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_engine.txt; : > $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_I8"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d gpurun_out/pmc_e$i -o e --output-format csv -- python tools/lead_time.py $W $H > gpurun_out/pmc_e$i.log 2>&1
  python - "$i" >> $out <<'PY'
import csv,glob,collections,sys
for f in glob.glob("gpurun_out/pmc_e%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "pl_engine" in r.get("Kernel_Name",""):
            agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in agg.items(): print(f"{k:28s} {v:16.0f}")
PY
done
cat $out
