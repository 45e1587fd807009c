"""Developer tool: print the kernel timeline of ONE factorisation from a rocprofv3 kernel-trace CSV."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('elfihip::', ''), r.get('Queue_Id', '?')))
rows.sort()
# last gram_kernel marks the last factorisation
idx = [i for i, r in enumerate(rows) if 'gram_kernel' in r[2]]
start = idx[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
t0 = rows[start][0]
n = 0
for s, e, name, q in rows[start:]:
    if 'kstar' in name:
        break
    print('%9.1f %9.1f  %7.1f us  q%-3s %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name[:40]))
    n += 1
    if n > int(sys.argv[3]) if len(sys.argv) > 3 else 60:
        break
