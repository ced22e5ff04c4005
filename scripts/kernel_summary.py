import csv, sys, glob, collections
import os
fn=max(glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv'), key=os.path.getmtime)
rows=list(csv.DictReader(open(fn)))
agg=collections.OrderedDict()
for r in rows:
    n=r['Kernel_Name']
    if not (n.startswith('k_') or n.startswith('void k_')): continue
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    key=(n.split('(')[0][:44], r['Grid_Size_X'],r['Grid_Size_Y'],r['Grid_Size_Z'])
    a=agg.setdefault(key,[0,0.0]); a[0]+=1; a[1]+=d
tot=sum(a[1] for a in agg.values())
print('total ms', tot/1e3)
t0=min(int(r['Start_Timestamp']) for r in rows); t1=max(int(r['End_Timestamp']) for r in rows)
print('first start to last end ms', (t1-t0)/1e6, ' launches', len(rows))
nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 28
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1])[:nrows]:
    print(f"{k[0]:44s} grid={k[1]:>10s},{k[2]:>5s},{k[3]:>4s} calls={a[0]:3d} total_ms={a[1]/1e3:8.2f} avg_us={a[1]/a[0]:9.1f} {100*a[1]/tot:5.1f}%")
