#!/bin/bash
# run on the GPU box: kernel trace + the two HBM counter passes for bench.py (summaries copied into profiles/)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 4 --no-cpu-baseline --no-e2e"      # (steady state: see bench.py --warmup)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
grep '^{"metric"' $OUT/trace.log > $OUT/trace_run.json || true      # the JSON line of the very run the kernel stats come from
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name '*.csv' | head -20
