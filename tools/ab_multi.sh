#!/bin/bash
# GPU box: the counting phase under several builds / settings in ONE run.  usage: tools/ab_multi.sh "<label>|<env assignments>" ...
cd "${GRAFT_REPO_ROOT:-.}"
GMB=${GMB:-47}; REPS=${REPS:-4}
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  echo "== $label  [$envs] =="
  env JASPER_COUNT_DEBUG=1 $envs python3 tools/bench_count_steps.py $GMB $REPS 2>&1 | grep 'abandoned\|^rep' | sed -e 's/.*its partition passes took/   abandoned: partition passes/' -e 's/ ms).*/ ms/' | sed -e 's/ -> .*stages/ stages/'
done
