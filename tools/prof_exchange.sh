#!/bin/bash
# GPU box: kernel trace of the multi-GPU counting exchange played on ONE GPU (tools/bench_exchange_steps.py), for W = 2, 4, 8 virtual
# ranks of the bench workload -> gpurun_out/xchg_prof/exchange_kernel_stats.csv (+ the tool's own stage lines)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/xchg_prof
rm -rf $OUT; mkdir -p $OUT
echo "W,Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs" > $OUT/exchange_kernel_stats.csv
for W in 2 4 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$W -- python3 tools/bench_exchange_steps.py $W 47 2 > $OUT/W$W.log 2>&1
  python3 - $OUT/t$W $W >> $OUT/exchange_kernel_stats.csv <<'PY'
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
fn = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True))[-1]
for r in csv.DictReader(open(fn)):
    n = r["Name"]; n = (n[5:] if n.startswith("void ") else n).split("(")[0]
    if n.startswith("jk::"):
        print(",".join([sys.argv[2], '"%s"' % n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]]))
PY
  grep -v amdgpu.ids $OUT/W$W.log | tail -5
  rm -rf $OUT/t$W
done
cat $OUT/exchange_kernel_stats.csv
