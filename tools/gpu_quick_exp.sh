#!/bin/bash
# like gpu_quick.sh, for the experiment library (libpngloss_hip_exp.so, tools/replay_clocks.sh with EXP_DEFS=...): usage bash tools/gpu_quick_exp.sh <tag>
export PNGLOSS_HIP_LIBNAME=libpngloss_hip_exp.so
exec bash $(dirname $0)/gpu_quick.sh "$@"
