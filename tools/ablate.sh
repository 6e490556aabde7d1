#!/bin/bash
# Timing-only ablation builds of the row engine (results are WRONG for PL_ABLATE>0; never shipped).
# usage (build container): bash tools/ablate.sh build ; (GPU box): bash tools/ablate.sh run
cd "$(dirname "$0")/../pngloss_amd/csrc"
if [ "$1" = build ]; then
  for a in 1 2 3 4 5 6; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPL_ABLATE=$a -c pl_engine.hip -o /tmp/pl_engine_ab$a.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libpngloss_hip_ab$a.so pl_prepost.o /tmp/pl_engine_ab$a.o pl_host.o
  done
else
  cd ../..
  for a in 0 1 2 3 4 5 6; do
    n=libpngloss_hip_ab$a.so; [ $a = 0 ] && n=libpngloss_hip.so
    echo "== ablate $a"; PNGLOSS_HIP_LIBNAME=$n python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); os.environ["PNGLOSS_HIP_DEBUG"]="1"
import pngloss_amd as P
img = P.synth_rgba(1920, 1080, 0, 0)
P.optimize_with_rows(img, 19, 2)
PY
  done
fi
