import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:48]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if "rolling" not in k and "long" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print('   %-24s n=%d avg=%.4g' % (c, len(v), sum(v)/len(v)))
