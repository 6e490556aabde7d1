#!/bin/bash
# Developer tool (build container): tools/build_inflate_variant.sh NAME "-DPLI_LOOP=1 ..." -> tools/ablate_build/libpngloss_hip_NAME.so
# (the product library with pl_inflate.hip compiled with extra -D flags; the other objects are the tree's)
set -e
cd "$(dirname "$0")/.."
NAME=$1; DEFS=${2:-}
B=tools/ablate_build/obj_$NAME
mkdir -p $B
S=pngloss_amd/csrc
make -C $S -s libpngloss_hip.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $DEFS -c $S/pl_inflate.hip -o $B/pl_inflate.o
OBJS=""
for f in pl_prepost pl_engine pl_rows pl_seg pl_pngread pl_emit pl_deflate pl_host; do OBJS="$OBJS $S/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ablate_build/libpngloss_hip_$NAME.so $OBJS $B/pl_inflate.o
echo built tools/ablate_build/libpngloss_hip_$NAME.so
