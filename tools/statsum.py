"""Print a rocprofv3 kernel_stats.csv (found under the given directory) as a short table."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print('%-72s calls %5s avg_us %10.1f pct %s' % (r['Name'][:72], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
