#!/bin/bash
# Developer tool (build container): the product library with pl_lead_asm.h regenerated under other generator settings and/or extra defines for pl_engine.hip:
#   tools/build_wg_variant.sh NAME "PL_LEAD_BURST=2 ..." "-DPL_LEAD_BURST0_CLEAN=2 ..."   -> tools/ablate_build/libpngloss_hip_NAME.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; GENENV=${2:-}; DEFS=${3:-}
T=tools/ablate_build/src_$NAME
rm -rf $T; mkdir -p $T
cp pngloss_amd/csrc/*.hip pngloss_amd/csrc/*.h pngloss_amd/csrc/*.c pngloss_amd/csrc/Makefile $T/
mkdir -p tools/ablate_build/include; cp include/pngloss_hip.h tools/ablate_build/include/ 2>/dev/null || true
env $GENENV python tools/gen_lead_asm.py > $T/pl_lead_asm.h
make -C pngloss_amd/csrc -s libpngloss_hip.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $DEFS -I include -c $T/pl_engine.hip -o $T/pl_engine.o
OBJS=""
for f in pl_prepost pl_rows pl_seg pl_pngread pl_inflate pl_emit pl_deflate pl_host; do OBJS="$OBJS pngloss_amd/csrc/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ablate_build/libpngloss_hip_$NAME.so $OBJS $T/pl_engine.o
echo built tools/ablate_build/libpngloss_hip_$NAME.so
