"""Developer tool: the kernel timeline of a few consecutive steps from a rocprofv3 --kernel-trace CSV.
usage: python scripts/step_timeline.py <kernel_trace.csv> <anchor kernel substring> [first anchor index] [anchors to show]
Prints, per kernel between the anchors: start offset, duration, gap to the previous kernel's end (us)."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                     r['Kernel_Name'].split('(')[0].replace('elfihip::', '')))
rows.sort()
anchor = sys.argv[2]
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
a0 = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) // 2
na = int(sys.argv[4]) if len(sys.argv) > 4 else 2
lo, hi = idx[a0], idx[min(a0 + na, len(idx) - 1)]
t0, prev = rows[lo][0], rows[lo][0]
for s, e, name in rows[lo:hi + 1]:
    print('%9.1f  %8.1f us  gap %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name[:70]))
    prev = e
