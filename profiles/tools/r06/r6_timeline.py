import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
first=sys.argv[2] if len(sys.argv)>2 else 'k_primary_mesh<false>'
idx=[i for i,r in enumerate(rows) if first in r['Kernel_Name']]
s=idx[-2]; e=idx[-1]
t0=int(rows[s]['Start_Timestamp'])
for r in rows[s:e]:
    n=r['Kernel_Name'].replace('void grt::(anonymous namespace)::','').replace('(grt::RenderArgs)','').replace('void grt::','')
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} .. {(int(r['End_Timestamp'])-t0)/1e3:9.1f} us  (+{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f})  {n[:60]}  grid {r.get('Grid_Size_X','?')}")
