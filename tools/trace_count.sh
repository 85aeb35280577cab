#!/bin/bash
# GPU box: kernel trace (start / end of every dispatch) of the counting phase alone; usage: tools/trace_count.sh <out dir under gpurun_out> [env assignments ...]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-trace}; shift
mkdir -p $OUT
for a in "$@"; do export "$a"; done
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 tools/bench_count_steps.py 47 3 > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if any(n in r["Kernel_Name"] for n in ("part1", "part2", "region_insert", "import3h", "clamp", "decide"))]
# the last counting call: everything after the last-but-one region_insert
ri = [i for i, r in enumerate(ks) if "region_insert" in r["Kernel_Name"]]
first = ri[-2] + 1 if len(ri) > 1 else 0
t0 = int(ks[first]["Start_Timestamp"])
with open(out + "/timeline.txt", "w") as fo:
    for r in ks[first:]:
        nm = r["Kernel_Name"].split("(")[0].replace("jk::", "")[:70]
        line = "%9.3f .. %9.3f ms  (%7.3f)  q%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Queue_Id", "?"), nm)
        print(line); fo.write(line + "\n")
PY
rm -rf $OUT/kt
