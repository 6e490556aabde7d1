#!/bin/bash
# round 6: streams per CU of the device inflate (a 4 KB input stage instead of 16 KB: 50 KB of shared memory a stream, three streams a CU): aggregate at 512 / 768 files
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}"
  for v in ${INFL_VARIANTS}; do
    for n in 1 512 768; do
      echo "## PNGLOSS_HIP_LIBNAME=../../tools/ablate_build/libpngloss_hip_$v.so READ_DISTINCT=8 python tests/tools/gpu_read_time.py $n 1280 720 16"
      PNGLOSS_HIP_LIBNAME=../../tools/ablate_build/libpngloss_hip_$v.so READ_DISTINCT=8 timeout 600 python tests/tools/gpu_read_time.py $n 1280 720 16 2>&1 | grep "library calls" | tail -2 | cut -c20-260
    done
  done
} > $OUT/r06_inflate_streams.txt 2>&1
