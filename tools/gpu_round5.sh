#!/bin/bash
# One GPU-box session for the round-5 code (gpu_round4.sh + stamps: every file says which sources it was taken at -- source_digest = pngloss_amd.source_digest(), head = $HEAD of the caller): parity tests, smoke, the default bench line, rocprofv3 kernel trace + PMC traffic passes
# (separate passes; counters only with --kernel-trace), LDS counters of the kernels that ship, phase clocks, host CPU of the asynchronous
# entry, coverage table of the segment engine, kernel trace of a seeded (strength, bleed) pair.  Every file gets a header that says what
# was run and whether a profiling switch slowed it.  Outputs -> gpurun_out/<tag>_*  (the builder copies what is to be judged to profiles/)
# usage: HEAD=<git sha> tools/gpu_round5.sh [tag] [parts]     parts: any of  tests bench trace pmc lds clocks async cover seeded fuzz   (default: all but fuzz)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05}
PARTS=${2:-"tests bench trace pmc clocks async seeded batch"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
DIGEST=$(python -c 'import pngloss_amd as P; print(P.source_digest())' 2>/dev/null)
STAMP="source_digest=$DIGEST head=${HEAD:-unknown}"
BOX="$STAMP; one MI355X (gpurun box), $(python -c 'import pngloss_amd as P; print(P.hip_lib().pngloss_hip_version().decode())' 2>/dev/null)"

if has tests; then
  { echo "# $STAMP"; ( time timeout 1500 python -m pytest tests -m gpu -q ) 2>&1; echo "pytest rc=$?"; } > $OUT/${TAG}_pytest_gpu.txt 2>&1
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/${TAG}_smoke.txt
fi
if has bench; then
  python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?" >> $OUT/${TAG}_bench.err
fi
BENCHARGS="--no-cpu-baseline --no-batch --no-sweep"
if has trace; then
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_trace -o trace --output-format csv -- python $R/bench.py --steps 2 --warmup 1 $BENCHARGS > $OUT/${TAG}_prof_trace.log 2>&1
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 $BENCHARGS   ($BOX; 3 engine runs of the 4096x4096 frame, s=19 b=2; no profiling switch of the library set)"; cat $(find $OUT/${TAG}_prof_trace -name "*kernel_stats.csv" | head -1); } > $OUT/${TAG}_kernel_trace_stats.txt
  rm -rf $OUT/${TAG}_prof_trace
fi
if has pmc; then
  for C in FETCH_SIZE WRITE_SIZE; do
    # (counter collection serialises the dispatches of all queues: the caller's stream must not hold a wait for the engine's "finished" word in front of them -- the blocking variant of the entry)
    PNGLOSS_HIP_NO_STREAM_WAIT=1 rocprofv3 --kernel-trace --pmc $C -d $OUT/${TAG}_prof_$C -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 $BENCHARGS > $OUT/${TAG}_prof_$C.log 2>&1
  done
  python - "$OUT" "$TAG" "$STAMP" > $OUT/${TAG}_pmc_fetch_write.txt <<'PY'
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
print("# " + (sys.argv[3] if len(sys.argv) > 3 else ""))
print("# PNGLOSS_HIP_NO_STREAM_WAIT=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch --no-sweep")
print("# summed over the dispatches of each kernel during ONE engine run of the 4096x4096 frame (s=19 b=2); raw counter units (KB; FETCH_SIZE counts 64 B per 128 B request on gfx950)")
print("%-28s %10s %16s %16s" % ("kernel", "dispatches", "FETCH_SIZE", "WRITE_SIZE"))
for n in sorted(agg, key=lambda k: -agg[k]["FETCH_SIZE"]):
    print("%-28s %10d %16.1f %16.1f" % (n[:28], calls[(n, "FETCH_SIZE")], agg[n]["FETCH_SIZE"], agg[n]["WRITE_SIZE"]))
seg = [n for n in agg if n.startswith("seg_k_")]
ef = sum(agg[n]["FETCH_SIZE"] for n in seg); ew = sum(agg[n]["WRITE_SIZE"] for n in seg)
res = {"round": tag, "stamp": sys.argv[3] if len(sys.argv) > 3 else "", "kernel": "segment-parallel row engine (seg_k_ctl [control + validation] + seg_k_enum + seg_k_chain + seg_k_replay, all dispatches of one engine run)",
       "workload": "4096x4096 RGBA8 s=19 b=2", "FETCH_SIZE_KB_raw": ef, "WRITE_SIZE_KB_raw": ew, "fetch_correction": 2.0, "write_correction": 1.0,
       "traffic_bytes": int((ef * 2.0 + ew) * 1024), "algorithmic_bytes": 8 * 4096 * 4096,
       "note": "FETCH_SIZE x2 (gfx950: 64 B per 128 B request, MI355X_MICROARCH.md; the same factor the 64 MiB copy calibrated in rounds 1-2). The engine keeps its working set (state maps, decision tables, candidate rows: a few MB per row attempt) in L2/MALL and re-reads it every row attempt: traffic is what reaches the memory side of L2, not the algorithmic 8 B/px"}
json.dump(res, open(f"{out}/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
  rm -rf $OUT/${TAG}_prof_FETCH_SIZE $OUT/${TAG}_prof_WRITE_SIZE
fi
if has lds; then
  : > $OUT/${TAG}_lds_raw.txt
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp -d $OUT/${TAG}_lds$i -o e --output-format csv -- python tests/tools/gpu_seg_time.py 4096 512 0 19 2 1 > $OUT/${TAG}_lds$i.log 2>&1
  done
  python - "$OUT" "$TAG" "$BOX" > $OUT/${TAG}_lds_pmc.txt <<'PY'
import csv, glob, collections, sys
out, tag, box = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(f"{out}/{tag}_lds*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "seg_k_" not in n or "resolve" in n: continue
        k = n.split("seg_k_")[1].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
cols = ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_INST_LEVEL_LDS"]
print("# rocprofv3 --kernel-trace --pmc <counters> -- python tests/tools/gpu_seg_time.py 4096 512 0 19 2 1   (%s; separate passes per counter group;" % box)
print("# averages per dispatch of each kernel of the segment engine AS IT SHIPS (seg_k_enum<512> for this width), summed over the device as rocprofv3 reports them; no profiling switch of the library set)")
print("%-12s" % "kernel" + "".join("%22s" % c for c in cols) + "%12s" % "LdsLatency")
for k in ("ctl", "enum<512>", "enum<1024>", "chain", "replay"):
    if k not in agg: continue
    v = {c: agg[k][c] / max(1, len(disp[(k, c)])) for c in cols}
    lat = v["SQ_INST_LEVEL_LDS"] / v["SQ_INSTS_LDS"] if v["SQ_INSTS_LDS"] else 0
    print("%-12s" % k + "".join("%22.0f" % v[c] for c in cols) + "%12.1f" % lat)
PY
  rm -rf $OUT/${TAG}_lds1 $OUT/${TAG}_lds2 $OUT/${TAG}_lds3
fi
if has clocks; then
  { echo "# PNGLOSS_HIP_SEGPROF=1 PNGLOSS_HIP_DEBUG=1 python tests/tools/gpu_seg_time.py 4096 4096 0 19 2 2   ($BOX)"
    echo "# phase clocks inside the kernels of the segment engine (100 MHz wall clock, per workgroup).  READING THE CLOCK DRAINS THE QUEUES: with this switch the engine"
    echo "# takes about TWICE as long as it ships (compare the Mpx/s below with profiles/${TAG}_bench.json) -- the figures give proportions inside a kernel, not its duration."
    PNGLOSS_HIP_SEGPROF=1 PNGLOSS_HIP_DEBUG=1 python tests/tools/gpu_seg_time.py 4096 4096 0 19 2 2 2>&1 | grep -v amdgpu.ids; } > $OUT/${TAG}_seg_phase_clocks.txt
fi
if has async; then
  { echo "# python tests/tools/gpu_async_check.py [W H s b]   ($BOX): pngloss_hip_optimize_batch_async -- time to return, host CPU time of the whole process"
    echo "# (getrusage RUSAGE_SELF: all threads, i.e. the context's launch thread included) between the call and the end of pngloss_hip_finish, 50 ms of host sleep in between"
    python tests/tools/gpu_async_check.py 2>&1 | grep -v amdgpu.ids
    python tests/tools/gpu_async_check.py 1920 1080 2>&1 | grep -v amdgpu.ids
    python tests/tools/gpu_async_check.py 8192 2048 85 2 2>&1 | grep -v amdgpu.ids; } > $OUT/${TAG}_host_cpu.txt
fi
if has cover; then
  python tests/tools/gpu_seg_coverage.py 1024 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_seg_coverage.txt
fi
if has seeded; then
  PNGLOSS_HIP_ENGINE=seg rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_seeded -o trace --output-format csv -- python tests/tools/gpu_seg_time.py 8192 1024 0 85 2 1 > $OUT/${TAG}_prof_seeded.log 2>&1
  { echo "# PNGLOSS_HIP_ENGINE=seg rocprofv3 --kernel-trace --stats -- python tests/tools/gpu_seg_time.py 8192 1024 0 85 2 1   ($BOX; an 8192x1024 strip of the configs[4] frame at s=85 b=2: the SEEDED enumeration)"; cat $(find $OUT/${TAG}_prof_seeded -name "*kernel_stats.csv" | head -1); } > $OUT/${TAG}_kernel_trace_stats_seeded.txt
  rm -rf $OUT/${TAG}_prof_seeded
fi
if has fuzz; then
  { echo "# randomised parity campaign against the CPU oracle (tests/tools/gpu_fuzz.py: random shapes, contents, strengths 0..255 incl. 85 / 100 / 127 / 200 / 255, bleeds 1..32767, both row_filters modes, device batches of 5 mixed images), row engine pinned case by case"
    for seed in 41 42; do
      echo "## FUZZ_ENGINES=seg,,seg,mix  python tests/tools/gpu_fuzz.py 150 $seed"; FUZZ_ENGINES=seg,,seg,mix timeout 400 python tests/tools/gpu_fuzz.py 150 $seed 2>&1 | grep -v amdgpu.ids | tail -3
      echo "## FUZZ_ENGINES=seg,  python tests/tools/gpu_fuzz.py 150 $seed big"; FUZZ_ENGINES=seg, timeout 400 python tests/tools/gpu_fuzz.py 150 $seed big 2>&1 | grep -v amdgpu.ids | tail -3
    done; } > $OUT/${TAG}_fuzz_campaign.txt
fi
if has batch; then
  # batches: a kernel timeline of 32 frames of 1080p (two launch groups), n-frame curves on the segment engine, small and mixed batches in one process
  bash tools/archive_r5/gpu_r5_timeline.sh 32 ${TAG}
  { echo "# $STAMP"; cat $OUT/${TAG}_timeline_32.txt; } > $OUT/${TAG}_batch_timeline_32.txt; rm -f $OUT/${TAG}_timeline_32.txt
  { echo "# $STAMP"; echo "# n frames of 1920x1080 (generator mode 0, s=19 b=2) in one device-resident batch, engine ms (best of 2) from the library's events: tests/tools/gpu_seg_batch.py ($BOX)"; python tests/tools/gpu_seg_batch.py 1920 1080 1 2 4 8 16 32 64 128 256 2>&1 | grep -v amdgpu.ids; } > $OUT/${TAG}_batch_curve.txt
  { echo "# $STAMP"; echo "# python tests/tools/gpu_small_batches.py 3 ($BOX)"; python tests/tools/gpu_small_batches.py 3 2>&1 | grep -v amdgpu.ids; } > $OUT/${TAG}_small_batches.txt
fi
ls -la $OUT | grep ${TAG}_ | awk '{print $5, $9}'
