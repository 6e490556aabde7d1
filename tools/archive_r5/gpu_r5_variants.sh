#!/bin/bash
# Developer tool (GPU box): 1080p batches on the segment engine for prebuilt variant libraries (tools/build_variant.sh), each with optional environment:
#   tools/gpu_r5_variants.sh TAG "NS" name[:ENV=V,ENV=V] ...      name "base" = the tree's library
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=$1; NS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd $R
cp pngloss_amd/csrc/libpngloss_hip.so /tmp/keep.so
: > $OUT/${TAG}_variants.txt
for V in "$@"; do
  name=${V%%:*}; envs=""; [ "$name" != "$V" ] && envs=$(echo "${V#*:}" | tr ',' ' ')
  if [ "$name" = base ]; then cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so; else cp tools/ablate_build/libpngloss_hip_$name.so pngloss_amd/csrc/libpngloss_hip.so; fi
  echo "=== $V" >> $OUT/${TAG}_variants.txt
  env $envs SEG_BATCH_ENGINES=seg timeout 900 python tests/tools/gpu_seg_batch.py ${VW:-1920} ${VH:-1080} $NS 2>&1 | grep -v "^---" | grep -v "^ *$" >> $OUT/${TAG}_variants.txt
done
cp /tmp/keep.so pngloss_amd/csrc/libpngloss_hip.so
