import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'prepare' in r['Kernel_Name']]
i0,i1=idx[-2],idx[-1]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1]:
    print(round((int(r['Start_Timestamp'])-t0)/1000,1), round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000,1), r['Kernel_Name'][:48])
