#!/bin/bash
# One quick look on the GPU box: the headline frame three times (engine ms) and the average duration of the engine's kernels (rocprofv3).
# usage (through gpurun): bash tools/gpu_quick.sh <tag>   -> gpurun_out/<tag>_time.txt
TAG=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tests/tools/gpu_seg_time.py 4096 4096 0 19 2 3 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_time.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_tr -o t --output-format csv -- python tests/tools/gpu_seg_time.py 4096 4096 0 19 2 2 > /dev/null 2>&1
python - "$(find gpurun_out/${TAG}_tr -name "*kernel_stats.csv" | head -1)" >> gpurun_out/${TAG}_time.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "seg_k_" in r["Name"] and "resolve" not in r["Name"]:
        print("%-14s calls %6s  avg %8.2f us" % (r["Name"].split("seg_k_")[1].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/${TAG}_tr
