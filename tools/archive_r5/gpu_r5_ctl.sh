#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
PNGLOSS_HIP_SEG_GROUPS=1 bash tools/gpu_r5_prof.sh 32 r05_g1
cp pngloss_amd/csrc/libpngloss_hip.so /tmp/keep.so
cp tools/ablate_build/libpngloss_hip_noval.so pngloss_amd/csrc/libpngloss_hip.so
PNGLOSS_HIP_SEG_GROUPS=1 bash tools/gpu_r5_prof.sh 32 r05_g1_noval
cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so
