#!/bin/bash
# GPU box: HBM byte counters of the counting kernels for one setting (two separate --pmc passes)
#   tools/pmc_count.sh <label> [ENV=VALUE ...]       -> gpurun_out/pmc_<label>.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
label=$1; shift
for a in "$@"; do export "$a"; done
OUT=gpurun_out/pmc_$label
mkdir -p $OUT
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 tools/bench_count_steps.py ${GMB:-47} 2 > $OUT/w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 tools/bench_count_steps.py ${GMB:-47} 2 > $OUT/f.log 2>&1
{ python3 tools/pmc_summary.py $OUT/w; python3 tools/pmc_summary.py $OUT/f; } | grep -A1 "part1\|part2\|region_insert" > gpurun_out/pmc_$label.txt
rm -rf $OUT
cat gpurun_out/pmc_$label.txt
