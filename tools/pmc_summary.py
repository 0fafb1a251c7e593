"""Summarise rocprofv3 --pmc csv output: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + '/*/*counter_collection.csv'):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[(r['Kernel_Name'][:50], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in sorted(agg.items()):
        print('%-52s %-12s n=%d mean=%.1f' % (k, c, len(v), sum(v) / len(v)))
