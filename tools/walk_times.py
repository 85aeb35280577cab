import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
out=[]
for r in rows:
    n=r['Kernel_Name']
    if any(x in n for x in ('seg_walk','scan_batch','rescan','stitch','classify','find_sync','seg_init')):
        out.append("%s=%.0f"%(n.split('::')[1].split('(')[0].replace('_kernel',''),(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
print(' '.join(out[-40:]))
