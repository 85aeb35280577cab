#!/bin/bash
for m in 0 1 2; do echo "MODE $m"; JASPER_EXPERIMENT_MODE=$m python tools/bench_count.py 47 29 2 2>&1 | grep rep; done
