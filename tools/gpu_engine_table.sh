#!/bin/bash
# GPU box: the figures of DESIGN.md section 6's round-3 table that bench.py does not print (other strengths, other image classes, the
# suite images, small batches), each with both row engines where that says something.  Output -> gpurun_out/<tag>_engine_table.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03}; OUT=$R/gpurun_out/${TAG}_engine_table.txt; mkdir -p $R/gpurun_out; cd $R
{
echo "# 4096x4096 mode 0, strength / bleed pairs, segment engine (default choice) and workgroup engine"
for sb in "19 2" "20 2" "20 1" "40 2" "85 8" "0 2" "7 3"; do
  python tests/tools/gpu_seg_time.py 4096 4096 0 $sb 2 2>&1 | tail -1
  PNGLOSS_HIP_ENGINE=wg python tests/tools/gpu_seg_time.py 4096 4096 0 $sb 1 2>&1 | tail -1 | sed 's/^/   workgroup engine: /'
done
echo "# one 1920x1080 frame, generator modes 0..5, s=19 b=2"
for m in 0 1 2 3 4 5; do python tests/tools/gpu_seg_time.py 1920 1080 $m 19 2 2 2>&1 | tail -1; done
echo "# 8192x8192 mode 0 s=19 b=2"
python tests/tools/gpu_seg_time.py 8192 8192 0 19 2 1 2>&1 | tail -1
echo "# the eleven suite images, one at a time, both engines"
python tests/tools/gpu_suite_time.py 2>&1 | grep -v "^ *$"
echo "# n frames in one device-resident batch, both engines"
python tests/tools/gpu_seg_batch.py 1920 1080 1 2 4 8 16 32 2>&1 | grep -v "^ *$"
python tests/tools/gpu_seg_batch.py 512 512 1 4 16 64 2>&1 | grep -v "^ *$"
} > $OUT 2>&1
cat $OUT
