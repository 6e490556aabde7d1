#!/bin/bash
# Developer tool (build container): tools/build_variant.sh NAME "-DSEG_UNIT=2 ..." -> tools/ablate_build/libpngloss_hip_NAME.so
# (the product library with extra -D flags; only the sources that include pl_seg_core.h are recompiled, the other objects are the tree's)
set -e
cd "$(dirname "$0")/.."
NAME=$1; DEFS=${2:-}
B=tools/ablate_build/obj_$NAME
mkdir -p $B
S=pngloss_amd/csrc
make -C $S -s libpngloss_hip.so
for f in pl_seg pl_host; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $DEFS -c $S/$f.hip -o $B/$f.o &
done
wait
OBJS=""
for f in pl_prepost pl_engine pl_rows pl_pngread pl_inflate pl_emit pl_deflate; do OBJS="$OBJS $S/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ablate_build/libpngloss_hip_$NAME.so $OBJS $B/pl_seg.o $B/pl_host.o
echo built tools/ablate_build/libpngloss_hip_$NAME.so
