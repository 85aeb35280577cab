#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/polish_prof2
rm -rf $OUT; mkdir -p $OUT
JASPER_POLISH_LANES=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 tools/bench_polish_steps.py 47 4 > $OUT/run.log 2>&1
grep "^rep" $OUT/run.log | cut -c1-60
python3 - $OUT/t <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
starts = [i for i, n in enumerate(names) if 'scan_classify' in n]
# calls: a call has ONE scan_classify (pass 0)
for ci, st in enumerate(starts):
    en = starts[ci + 1] if ci + 1 < len(starts) else len(rows)
    t0 = int(rows[st]['Start_Timestamp']); tot = 0; per = {}
    for r in rows[st:en]:
        n = r['Kernel_Name']; n = (n[5:] if n.startswith('void ') else n).split('(')[0].replace('jk::', '')
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); tot += d; per[n] = per.get(n, 0) + d
    span = int(rows[en - 1]['End_Timestamp']) - t0
    print("call %d: %d kernels, span %.2f ms, kernel sum %.2f ms; " % (ci, en - st, span / 1e6, tot / 1e6) + ", ".join("%s %.2f" % (k.replace('_kernel',''), v / 1e6) for k, v in sorted(per.items(), key=lambda x: -x[1])[:6]))
PY
