import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'ptr::' not in n: continue
        short = n.split('(')[0].replace('void ', '')
        agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
for kname, cs in agg.items():
    print(kname)
    for c, vals in sorted(cs.items()):
        print(f"   {c:28s} {sum(vals)/len(vals):16.0f}  (n={len(vals)})")
