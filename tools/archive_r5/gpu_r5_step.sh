#!/bin/bash
# Developer tool (GPU box): a round-5 iteration step -- the GPU tests that exercise batches on the segment engine, then batch timings (tree's library, optional env variants)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r05s}; shift
cd /tmp && export TMPDIR=/tmp
cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "configs3 or segment_engine or mixed_images or digest_headline or suite_batch or two_contexts" 2>&1 | tail -5 ) > $OUT/${TAG}_tests.txt 2>&1
bash tools/gpu_r5_variants.sh $TAG "${NS:-16 32 64 128}" "$@"
