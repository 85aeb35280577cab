#!/bin/bash
# GPU box: kernel trace of one polish call of the bench workload (the last of 3): every kernel's duration in launch order
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/polish_prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 tools/bench_polish_steps.py 47 3 > $OUT/run.log 2>&1
grep "^rep" $OUT/run.log
python3 - $OUT/t <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last call: from the last scan_classify on
names = [r['Kernel_Name'] for r in rows]
start = max(i for i, n in enumerate(names) if 'scan_classify' in n)
t0 = int(rows[start]['Start_Timestamp']); prev_end = t0
tot = 0
for r in rows[start:]:
    n = r['Kernel_Name']; n = (n[5:] if n.startswith('void ') else n).split('(')[0].replace('jk::', '').replace('_kernel', '')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f us  +%6.1f gap  %-28s %7.1f us" % ((s - t0) / 1e3, (s - prev_end) / 1e3, n[:28], (e - s) / 1e3))
    prev_end = e; tot += e - s
print("kernels %.1f us, span %.1f us" % (tot / 1e3, (prev_end - t0) / 1e3))
PY
rm -rf $OUT/t
