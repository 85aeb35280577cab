#!/usr/bin/env python3
"""sum rocprofv3 --pmc counter_collection.csv per kernel:  python tools/pmc_summary.py <dir> [name filter]"""
import csv, glob, os, sys, collections
csv.field_size_limit(1 << 30)
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "jk::"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for fn in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        name = r["Kernel_Name"]
        name = (name[5:] if name.startswith("void ") else name).split("(")[0]
        if flt not in name:
            continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (name, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            calls[name] += 1
for name in sorted(acc):
    print(name, "calls", calls[name])
    for c, v in sorted(acc[name].items()):
        print("   %-28s %.4g total  %.4g per call" % (c, v, v / max(calls[name], 1)))
