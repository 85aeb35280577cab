#!/bin/bash
# GPU box: cProfile of the drop-in's main thread on the configs[1] files (written to /tmp/e2e_prof)
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
mkdir -p /tmp/e2e_prof && cd /tmp/e2e_prof
[ -f reads.fq ] || PYTHONPATH=$R python3 -c "from jasper_amd import synth; synth.write_cli_inputs('.', 47.0, 2, coverage=30)"
for f in *; do case $f in reads.fq|asm.fa) ;; *) rm -f "$f";; esac; done
PYTHONPATH=$R JASPER_AMD_NO_JF=1 JASPER_AMD_SLOW_EXIT=1 python3 -c "
import cProfile, pstats, sys
sys.argv = ['cli', '-r', 'reads.fq', '-a', 'asm.fa', '-k', '37', '-t', '16', '-p', '2']
from jasper_amd import cli
pr = cProfile.Profile()
pr.enable()
try:
    cli.run(sys.argv[1:])
finally:
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
" 2>&1 | cut -c1-160
