"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, counters averaged per dispatch."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:48]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[k][r['Counter_Name']] += 1
sel = sys.argv[2:] or ['k_f', 'thin']
for k, v in agg.items():
    if any(x in k for x in sel):
        print(k)
        for c in sorted(v):
            print('    {:32s} {:>16,.0f}'.format(c, v[c] / cnt[k][c]))
