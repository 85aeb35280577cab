#!/bin/bash
# GPU box: the counting phase with two builds of the library in ONE run (box-to-box differences are +-3 %)
#   tools/ab_count.sh <other .so> [genome_mb] [reps]
cd "${GRAFT_REPO_ROOT:-.}"
OTHER=$1; GMB=${2:-47}; REPS=${3:-4}
for i in 1 2; do
  echo "== A (in-tree build) =="; python3 tools/bench_count_steps.py $GMB $REPS | tail -n $((REPS-1))
  echo "== B ($OTHER) =="; JASPER_AMD_LIB=$OTHER python3 tools/bench_count_steps.py $GMB $REPS | tail -n $((REPS-1))
done
