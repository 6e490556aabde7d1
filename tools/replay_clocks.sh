#!/bin/bash
# Experiment build: the replay kernel's phase clocks (staging / walk / counts / sums + bound) in the enumeration's slots of the result record.
# Builds pngloss_amd/csrc/libpngloss_hip_exp.so (git-ignored) next to the product library; run on the GPU box with
#   PNGLOSS_HIP_LIBNAME=libpngloss_hip_exp.so PNGLOSS_HIP_SEGPROF=1 PNGLOSS_HIP_DEBUG=1 python tests/tools/gpu_seg_time.py 4096 4096 0 19 2 1
# and read the line "enumeration workgroups": load = staging, first steps = walk, remaining = counts, map = sums + bound (group 3 of every candidate).
set -e
cd $(dirname $0)/../pngloss_amd/csrc
mkdir -p /tmp/exp_build
for f in pl_prepost pl_engine pl_seg pl_pngread pl_inflate pl_emit pl_deflate pl_host; do
  if [ $f = pl_seg ]; then /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 ${EXP_DEFS:--DSEG_EXPERIMENT_REPLAY_CLOCKS=1} -c $f.hip -o /tmp/exp_build/$f.o; else cp $f.o /tmp/exp_build/$f.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ${EXP_LIB:-libpngloss_hip_exp.so} /tmp/exp_build/*.o
ls -la ${EXP_LIB:-libpngloss_hip_exp.so}
