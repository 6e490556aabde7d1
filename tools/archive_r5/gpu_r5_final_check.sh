#!/bin/bash
# Developer tool (GPU box): the round's last look at the tree as it stands -- GPU tests, smoke, default bench line, a short parity campaign with fresh seeds.
# usage: HEAD=<git sha> tools/gpu_r5_final_check.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05z}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
DIGEST=$(python -c 'import pngloss_amd as P; print(P.source_digest())' 2>/dev/null)
STAMP="source_digest=$DIGEST head=${HEAD:-unknown}"
{ echo "# $STAMP"; ( time timeout 1500 python -m pytest tests -m gpu -q ) 2>&1 | tail -12; } > $OUT/${TAG}_pytest_gpu.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/${TAG}_smoke.txt
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?" >> $OUT/${TAG}_bench.err
{ echo "# $STAMP"
  for seed in 51 52; do
    echo "## FUZZ_ENGINES=seg,,seg,mix  python tests/tools/gpu_fuzz.py 120 $seed"; FUZZ_ENGINES=seg,,seg,mix timeout 400 python tests/tools/gpu_fuzz.py 120 $seed 2>&1 | grep -v amdgpu.ids | tail -2
    echo "## FUZZ_ENGINES=seg,  python tests/tools/gpu_fuzz.py 120 $seed big"; FUZZ_ENGINES=seg, timeout 400 python tests/tools/gpu_fuzz.py 120 $seed big 2>&1 | grep -v amdgpu.ids | tail -2
  done; } > $OUT/${TAG}_fuzz.txt 2>&1
