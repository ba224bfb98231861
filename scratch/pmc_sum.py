import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
pat=sys.argv[2] if len(sys.argv)>2 else 'igemm2'
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    if pat in r['Kernel_Name']:
        agg[r['Kernel_Name'][:50]][r['Counter_Name']]+=float(r['Counter_Value']); 
disp=collections.defaultdict(set)
for r in rows:
    if pat in r['Kernel_Name']: disp[r['Kernel_Name'][:50]].add(r['Dispatch_Id'])
for k,v in agg.items():
    n=len(disp[k]); print(k,'dispatches',n)
    for c,val in sorted(v.items()): print('   %-32s %16.0f per dispatch'%(c,val/n))
