#!/bin/bash
# run on the GPU box: kernel trace of the list exchange played on one GPU (tools/bench_exchange_steps.py, 8 virtual ranks)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_xchg
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_exchange_steps.py 8 47 3 > $OUT/trace.log 2>&1
find $OUT -name '*kernel_stats.csv' | head -3
