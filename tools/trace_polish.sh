#!/bin/bash
# GPU box: kernel trace of one steady-state polish call (all lanes), one line per dispatch with its queue; usage: tools/trace_polish.sh <dir under gpurun_out> [env ...]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-trace_polish}; shift
mkdir -p $OUT
for a in "$@"; do export "$a"; done
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/kt -- python3 tools/bench_polish_steps.py 47 6 > $OUT/run.log 2>&1
grep "^rep" $OUT/run.log | cut -c1-60
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "scan_classify" in n]
# calls: groups of scan_classify dispatches closer than 2 ms to each other
calls = []
for i in starts:
    if calls and int(rows[i]["Start_Timestamp"]) - int(rows[calls[-1][-1]]["Start_Timestamp"]) < 2e6: calls[-1].append(i)
    else: calls.append([i])
first = calls[-1][0]
t0 = int(rows[first]["Start_Timestamp"])
queues = sorted({r["Queue_Id"] for r in rows[first:]})
with open(out + "/timeline.txt", "w") as fo:
    for r in rows[first:]:
        nm = r["Kernel_Name"].split("(")[0].replace("jk::", "").replace("void ", "").replace("_kernel", "")[:34]
        q = queues.index(r["Queue_Id"])
        line = "%9.1f .. %9.1f us (%7.1f)  %s%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "                         " * q, nm)
        print(line); fo.write(line + "\n")
PY
rm -rf $OUT/kt
