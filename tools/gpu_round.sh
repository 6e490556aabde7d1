#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel trace + PMC traffic passes.  Outputs -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-run}
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/${TAG}_smoke.txt
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?" >> $OUT/${TAG}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_trace -o trace -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch > $OUT/${TAG}_prof_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_prof_fetch -o fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch > $OUT/${TAG}_prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_prof_write -o write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch > $OUT/${TAG}_prof_write.log 2>&1
cd $R
python tools/rocpd_summary.py $OUT/${TAG}_prof_trace/trace_results.db > $OUT/${TAG}_kernel_trace_stats.txt 2>&1
python tools/rocpd_summary.py $OUT/${TAG}_prof_fetch/fetch_results.db $OUT/${TAG}_prof_write/write_results.db > $OUT/${TAG}_pmc_fetch_write.txt 2>&1
tools/pmc_engine.sh 4096 1024 > /dev/null 2>&1; cp $OUT/pmc_engine.txt $OUT/${TAG}_engine_sq.txt; rm -rf $OUT/pmc_e[0-9]*
rm -rf $OUT/${TAG}_prof_trace $OUT/${TAG}_prof_fetch $OUT/${TAG}_prof_write
tail -3 $OUT/${TAG}_pytest_gpu.txt; tail -2 $OUT/${TAG}_smoke.txt; cat $OUT/${TAG}_bench.json
