#!/bin/bash
# round 5: the new GPU tests + the default bench line (with the new legs)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "configs3 or wraps or one_damaged or rejects" > $OUT/r05_check_tests.txt 2>&1; echo "pytest rc=$?" >> $OUT/r05_check_tests.txt
( time python bench.py > $OUT/r05_check_bench.json 2> $OUT/r05_check_bench.err ) 2>> $OUT/r05_check_bench.err; echo "bench rc=$?" >> $OUT/r05_check_bench.err
