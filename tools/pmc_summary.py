import csv, glob, collections, os, sys
root = sys.argv[1]
for d in sorted(glob.glob(root + '/*set*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if 'spmm' not in k and 'spmv' not in k: continue
            agg[(k.split('(')[0][-42:], row['Counter_Name'])].append(float(row['Counter_Value']))
        for (k,c),vals in sorted(agg.items()):
            print(f'{os.path.basename(d):14s} {k:42s} {c:28s} n={len(vals)} mean={sum(vals)/len(vals):.4g}')
