"""Sum rocprofv3 --pmc counter_collection csv per kernel name: python tools/pmcsum.py DIR"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (k, r['Dispatch_Id'])
        if key not in seen:
            seen.add(key); n[k] += 1
for k, d in acc.items():
    if 'aspire' not in k: continue
    print(k, 'dispatches', n[k])
    for c, v in sorted(d.items()):
        print('   %-28s %16.0f per dispatch %14.0f' % (c, v, v / n[k]))
