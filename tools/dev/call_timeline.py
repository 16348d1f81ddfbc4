import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
mark=sys.argv[2]
idx=[i for i,r in enumerate(rows) if mark in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
t0=int(rows[a]['Start_Timestamp']); 
print(b-a,"launches; span %.1f us"%((int(rows[b]['Start_Timestamp'])-t0)/1e3))
for r in rows[a:b]:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:64]
    print("%8.1f %7.1f  %s"%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,n))
