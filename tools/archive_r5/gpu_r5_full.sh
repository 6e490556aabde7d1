#!/bin/bash
# round 5: the whole GPU suite + smoke + the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/${TAG}_smoke.txt
( time python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err ) 2>> $OUT/${TAG}_bench.err; echo "bench rc=$?" >> $OUT/${TAG}_bench.err
