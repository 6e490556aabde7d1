#!/bin/bash
# round 6: randomised parity campaigns against the CPU oracle (tests/tools/gpu_fuzz.py), the library's own choices and the seeds paths pinned
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  echo "# randomised parity campaign against the CPU oracle (tests/tools/gpu_fuzz.py: random shapes, contents, strengths 0..255, bleeds 1..32767, both row_filters modes, device batches of 5 mixed images)"
  for seed in ${FUZZ_SEEDS:-61 62}; do
    echo "## FUZZ_ENGINES=seg,,seg,mix  python tests/tools/gpu_fuzz.py 150 $seed"; FUZZ_ENGINES=seg,,seg,mix timeout 400 python tests/tools/gpu_fuzz.py 150 $seed 2>&1 | grep -v amdgpu.ids | tail -3
    echo "## FUZZ_ENGINES=seg,  python tests/tools/gpu_fuzz.py 120 $seed big"; FUZZ_ENGINES=seg, timeout 400 python tests/tools/gpu_fuzz.py 120 $seed big 2>&1 | grep -v amdgpu.ids | tail -3
    echo "## segments from seeds pinned (PNGLOSS_HIP_SEG_UNIT=0 PNGLOSS_HIP_SEG_SEEDS1=1)  FUZZ_ENGINES=seg python tests/tools/gpu_fuzz.py 150 $seed"; PNGLOSS_HIP_SEG_UNIT=0 PNGLOSS_HIP_SEG_SEEDS1=1 FUZZ_ENGINES=seg timeout 400 python tests/tools/gpu_fuzz.py 150 $seed 2>&1 | grep -v amdgpu.ids | tail -3
    echo "## ... big"; PNGLOSS_HIP_SEG_UNIT=0 PNGLOSS_HIP_SEG_SEEDS1=1 FUZZ_ENGINES=seg timeout 400 python tests/tools/gpu_fuzz.py 120 $seed big 2>&1 | grep -v amdgpu.ids | tail -3
    echo "## units from seeds pinned (PNGLOSS_HIP_SEG_UNIT=1)  FUZZ_ENGINES=seg python tests/tools/gpu_fuzz.py 150 $seed"; PNGLOSS_HIP_SEG_UNIT=1 FUZZ_ENGINES=seg timeout 400 python tests/tools/gpu_fuzz.py 150 $seed 2>&1 | grep -v amdgpu.ids | tail -3
    echo "## ... big"; PNGLOSS_HIP_SEG_UNIT=1 FUZZ_ENGINES=seg timeout 400 python tests/tools/gpu_fuzz.py 120 $seed big 2>&1 | grep -v amdgpu.ids | tail -3
    echo "## strength 19, bleed 2 only (every case has a seed set), segments from seeds"; FUZZ_STRENGTH=19 PNGLOSS_HIP_SEG_UNIT=0 PNGLOSS_HIP_SEG_SEEDS1=1 FUZZ_ENGINES=seg timeout 400 python tests/tools/gpu_fuzz.py 120 $seed 2>&1 | grep -v amdgpu.ids | tail -3
  done
} > $OUT/r06_fuzz_campaign.txt 2>&1
