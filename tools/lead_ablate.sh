#!/bin/bash
# Timing-only ablations of the band-leader inner loops (results are WRONG for n > 0): builds tools/ablate_build/libpngloss_hip_A<n>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ablate_build
for n in 1 2 3 4; do
  case $n in 1) export PL_LEAD_PREF=shadow;; 2) export PL_LEAD_PREF=start;; 3) export PL_LEAD_PREF=shadow PL_LEAD_BURST=8;; 4) export PL_LEAD_PREF=shadow PL_LEAD_BURST=2;; esac
  python tools/gen_lead_asm.py > tools/ablate_build/pl_lead_asm.h
  mkdir -p tools/ablate_build/src$n
  cp pngloss_amd/csrc/*.hip pngloss_amd/csrc/*.h tools/ablate_build/src$n/
  cp tools/ablate_build/pl_lead_asm.h tools/ablate_build/src$n/pl_lead_asm.h
  ( cd tools/ablate_build/src$n && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../../include -c pl_engine.hip -o pl_engine.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpngloss_hip_A$n.so pl_engine.o ../../../pngloss_amd/csrc/pl_prepost.o ../../../pngloss_amd/csrc/pl_emit.o ../../../pngloss_amd/csrc/pl_deflate.o ../../../pngloss_amd/csrc/pl_host.o ) &
done
wait
rm -rf tools/ablate_build/src* tools/ablate_build/pl_lead_asm.h
ls -la tools/ablate_build
