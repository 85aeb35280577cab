#!/usr/bin/env python3
"""kernel trace of the polish phase by stream (GPU box, after rocprofv3 --kernel-trace of tools/bench_polish_steps.py): start/end in us relative to the last scan_batch"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
scans = [i for i, r in enumerate(rows) if 'scan_classify_batch_kernel' in r['Kernel_Name']]
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
i0 = scans[-nl]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    n = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d < 15:
        continue
    print("%8.0f %8.0f  q%s  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r.get('Queue_Id', '?'), n.split('::')[1].split('(')[0]))
