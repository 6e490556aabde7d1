#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel trace + PMC traffic passes.  Outputs -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt
python bench.py > $OUT/bench.txt 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o trace -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -o write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_write.log 2>&1
cd $R
find $OUT -name "*.csv" | head -30
tail -3 $OUT/pytest_gpu.txt; cat $OUT/smoke.txt | tail -2; cat $OUT/bench.txt
