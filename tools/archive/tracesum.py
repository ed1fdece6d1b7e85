"""Group a rocprofv3 kernel_trace.csv by (kernel name, grid size): mean duration per group.
usage: python tools/tracesum.py DIR"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name']
    name = name.split('(anonymous namespace)::')[-1].split('(')[0] if 'aspire' in name else name[:40]
    grid = (r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'))
    acc[(name, grid)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('%-44s grid %-22s calls %5d mean_us %9.1f total_ms %8.2f' % (name[:44], grid, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
