#!/bin/bash
# round 6: one leg of the final campaign (seed 72, segments from seeds pinned) ended without its result line after ~400 s (its timeout); the same leg again finished normally
# (profiles/r06_fuzz72.txt).  This runs that configuration over more seeds with a per-case watchdog that names the case and the Python stack of a call that does not return.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  for seed in ${FUZZ_SEEDS:-72 72 73 74 75 76}; do
    echo "## PNGLOSS_HIP_SEG_UNIT=0 PNGLOSS_HIP_SEG_SEEDS1=1 FUZZ_ENGINES=seg FUZZ_WATCHDOG=45 timeout 300 python tests/tools/gpu_fuzz.py ${FUZZ_SECS:-150} $seed"
    PNGLOSS_HIP_SEG_UNIT=0 PNGLOSS_HIP_SEG_SEEDS1=1 FUZZ_ENGINES=seg FUZZ_WATCHDOG=45 timeout 300 python tests/tools/gpu_fuzz.py ${FUZZ_SECS:-150} $seed 2>&1 | grep -v amdgpu.ids | tail -40
    echo "rc=${PIPESTATUS[0]} last case: $(cat $OUT/fuzz_current_case.txt 2>/dev/null)"
  done
} > $OUT/r06_fuzz_hunt.txt 2>&1
