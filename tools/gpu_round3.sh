#!/bin/bash
# One GPU-box session for the round-3 engine: parity tests, smoke, bench, rocprofv3 kernel trace + PMC traffic passes (separate
# passes, counters only with --kernel-trace), phase clocks of the segment engine.  Outputs -> gpurun_out/<tag>_*
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r03}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/${TAG}_smoke.txt
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?" >> $OUT/${TAG}_bench.err
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_trace -o trace --output-format csv -- $BENCH > $OUT/${TAG}_prof_trace.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch   (one MI355X; 3 engine runs of the 4096x4096 frame)"; cat $(find $OUT/${TAG}_prof_trace -name "*kernel_stats.csv" | head -1); } > $OUT/${TAG}_kernel_trace_stats.txt
BENCH1="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${TAG}_prof_$C -o pmc --output-format csv -- $BENCH1 > $OUT/${TAG}_prof_$C.log 2>&1
done
python - "$OUT" "$TAG" > $OUT/${TAG}_pmc_fetch_write.txt <<'PY'
import csv, glob, collections, json, sys
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(int)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{tag}_prof_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].strip()
            if n.startswith("void "): n = n[5:]
            agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == c: calls[(n, c)] += 1
print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch")
print("# summed over the dispatches of each kernel during ONE engine run of the 4096x4096 frame; raw counter units (KB; FETCH_SIZE counts 64 B per 128 B request on gfx950)")
print("%-28s %10s %16s %16s" % ("kernel", "dispatches", "FETCH_SIZE", "WRITE_SIZE"))
for n in sorted(agg, key=lambda k: -agg[k]["FETCH_SIZE"]):
    print("%-28s %10d %16.1f %16.1f" % (n[:28], calls[(n, "FETCH_SIZE")], agg[n]["FETCH_SIZE"], agg[n]["WRITE_SIZE"]))
copy = [n for n in agg if "copyBuffer" in n]
fcorr = wcorr = None
if copy:
    # the 64 MiB frame clone bench.py makes before the timed region: known byte count -> calibrates both counters
    cf = agg[copy[0]]["FETCH_SIZE"] / max(1, calls[(copy[0], "FETCH_SIZE")]); cw = agg[copy[0]]["WRITE_SIZE"] / max(1, calls[(copy[0], "WRITE_SIZE")])
seg = [n for n in agg if n.startswith("seg_k_")]
ef = sum(agg[n]["FETCH_SIZE"] for n in seg); ew = sum(agg[n]["WRITE_SIZE"] for n in seg)
res = {"round": tag, "kernel": "segment-parallel row engine (seg_k_ctl + seg_k_enum + seg_k_chain + seg_k_replay + seg_k_post, all dispatches of one engine run)",
       "workload": "4096x4096 RGBA8 s=19 b=2", "FETCH_SIZE_KB_raw": ef, "WRITE_SIZE_KB_raw": ew, "fetch_correction": 2.0, "write_correction": 1.0,
       "traffic_bytes": int((ef * 2.0 + ew) * 1024), "algorithmic_bytes": 8 * 4096 * 4096,
       "note": "FETCH_SIZE x2 (gfx950: 64 B per 128 B request, MI355X_MICROARCH.md; the same factor the 64 MiB copy calibrated in rounds 1-2). The engine keeps its working set (state maps, decision tables, candidate rows: ~3 MB per row attempt) in L2/MALL and re-reads it every row attempt: traffic is what reaches the memory side of L2, not the algorithmic 8 B/px"}
json.dump(res, open(f"{out}/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
PNGLOSS_HIP_SEGPROF=1 PNGLOSS_HIP_DEBUG=1 python tests/tools/gpu_seg_time.py 4096 4096 0 19 2 2 > $OUT/${TAG}_seg_phase_clocks.txt 2>&1
rm -rf $OUT/${TAG}_prof_trace $OUT/${TAG}_prof_FETCH_SIZE $OUT/${TAG}_prof_WRITE_SIZE
tail -3 $OUT/${TAG}_pytest_gpu.txt; tail -2 $OUT/${TAG}_smoke.txt; head -c 600 $OUT/${TAG}_bench.json; echo; head -12 $OUT/${TAG}_kernel_trace_stats.txt | cut -c1-150; tail -25 $OUT/${TAG}_pmc_fetch_write.txt; cat $OUT/${TAG}_seg_phase_clocks.txt | tail -6
