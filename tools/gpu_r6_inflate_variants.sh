#!/bin/bash
# round 6: build variants of the device inflate (tools/build_inflate_variant.sh): one 1280x720 generator file and the suite's files, MB/s per stream
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  for v in ${INFL_VARIANTS}; do
    echo "## PNGLOSS_HIP_LIBNAME=../../tools/ablate_build/libpngloss_hip_$v.so python tests/tools/gpu_read_time.py 1 1280 720 16"
    PNGLOSS_HIP_LIBNAME=../../tools/ablate_build/libpngloss_hip_$v.so timeout 300 python tests/tools/gpu_read_time.py 1 1280 720 16 2>&1 | grep "library calls" | tail -1 | cut -c90-260
    echo "## ... python tests/tools/gpu_inflate_suite.py"
    PNGLOSS_HIP_LIBNAME=../../tools/ablate_build/libpngloss_hip_$v.so timeout 300 python tests/tools/gpu_inflate_suite.py 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-20,98-130
  done
} > $OUT/r06_inflate_variants.txt 2>&1
