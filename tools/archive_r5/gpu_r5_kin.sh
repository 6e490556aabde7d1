cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
: > gpurun_out/r05at_kin.txt
for K in 16 20 24 28 32; do
  for SB in "40 1" "85 1" "85 2"; do
    echo "kin=$K $(PNGLOSS_HIP_KIN=$K PNGLOSS_HIP_DEBUG=1 python tests/tools/gpu_seg_time.py 8192 1024 0 $SB 2 2>&1 | grep -E "engine [0-9.]+ ms =|walked step" | sed 's/.*candidate none dropped.*times, \([0-9]*\) segments walked.*/walked \1/' | sed 's/out=.*//' | tail -2 | tr '\n' ' ')" >> gpurun_out/r05at_kin.txt
  done
done
