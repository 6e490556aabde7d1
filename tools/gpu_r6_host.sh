#!/bin/bash
# round 6: the host side -- calibration of the engine cost model (what the reference box measures), launch thread pinned / not pinned (one 4096x4096 frame; eight contexts on one device)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
{ echo "# $(python -c 'import pngloss_amd as P; print("source_digest=" + P.source_digest())') head=${HEAD:-unknown}; $(nproc) CPUs visible, cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
  echo "## engine calibration (PNGLOSS_HIP_DEBUG=1, five processes, a batch of two 640x64 frames each: the first batch of a process calibrates)"
  for i in 1 2 3 4 5; do PNGLOSS_HIP_DEBUG=1 python tests/tools/gpu_rank_share.py 2 2>&1 | grep "engine calibration"; done
  echo "## one 4096x4096 frame, s=19 b=2, engine ms (tests/tools/gpu_seg_time.py ... 4 runs), launch thread pinned (default) / not pinned"
  for pin in default 0 default 0; do
    if [ $pin = 0 ]; then PNGLOSS_HIP_PIN=0 python tests/tools/gpu_seg_time.py 4096 4096 0 19 2 4 2>&1 | grep Mpx | sort -t' ' -k6 -n | head -1 | sed "s/^/PIN=0        /"
    else python tests/tools/gpu_seg_time.py 4096 4096 0 19 2 4 2>&1 | grep Mpx | sort -t' ' -k6 -n | head -1 | sed "s/^/PIN=default  /"; fi
  done
  echo "## configs[3] on eight contexts of one device (tests/tools/gpu_multi8.py 256 2)"
  for pin in default 0 default 0; do
    if [ $pin = 0 ]; then PNGLOSS_HIP_PIN=0 python tests/tools/gpu_multi8.py 256 2 2>&1 | grep -v amdgpu.ids | tail -1; else python tests/tools/gpu_multi8.py 256 2 2>&1 | grep -v amdgpu.ids | tail -1; fi
  done
  echo "## the same with the launch threads competing for TWO CPUs (taskset -c 0-1 ...; pinning cannot apply: the set is smaller than four): what starving them costs"
  taskset -c 0-1 python tests/tools/gpu_multi8.py 256 1 2>&1 | grep -v amdgpu.ids | tail -1
} > $OUT/r06_host_side.txt 2>&1
