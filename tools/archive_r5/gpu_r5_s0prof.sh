cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/s0prof -o t --output-format csv -- python tests/tools/gpu_s0_time.py > /dev/null 2>&1
grep -E "pl_rows|pl_hist|pl_classify|pl_rank|pl_init|pl_unpack|pl_repack" $(find gpurun_out/s0prof -name "*kernel_stats.csv" | head -1) | cut -c1-160 > gpurun_out/r05ar_s0prof.txt
python - >> gpurun_out/r05ar_s0prof.txt <<'PY'
import csv, glob
f = glob.glob("gpurun_out/s0prof/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "pl_rows" in r["Kernel_Name"]:
        print(r["Kernel_Name"].split("(")[0][-20:], r["Grid_Size_X"], r["Grid_Size_Y"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
rm -rf gpurun_out/s0prof
