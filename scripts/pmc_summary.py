"""Developer tool: per-kernel sums of rocprofv3 --pmc counters from the counter_collection CSV.
usage: python scripts/pmc_summary.py <counter_collection.csv> [kernel-substring]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r['Kernel_Name'].split('(')[0].replace('elfihip::', '').replace('void ', '')
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        acc[name][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[name].add(r['Dispatch_Id'])
for name in acc:
    n = len(cnt[name])
    print("%s  (%d dispatches)" % (name[:60], n))
    for c, v in sorted(acc[name].items()):
        print("    %-28s %16.0f   per dispatch %14.1f" % (c, v, v / n))
