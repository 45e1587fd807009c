"""Developer tool: gaps between the kernels of the acquisition lock-step from a rocprofv3 kernel-trace CSV.
usage: python scripts/gap_acq.py <kernel_trace.csv>"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('elfihip::', '').replace('void ', '')))
rows.sort()
names = ['kstar_kernel', 'tri_apply_kernel<0>', 'tri_reduce_kernel', 'tri_apply_kernel<1>', 'grad_kernel', 'finish_kernel']
dur = {n: [] for n in names}
gap = {n: [] for n in names}     # idle time BEFORE the kernel
steps = []
for i in range(1, len(rows)):
    s, e, n = rows[i]
    key = next((k for k in names if n.startswith(k)), None)
    if key is None:
        continue
    dur[key].append((e - s) / 1e3)
    gap[key].append((s - rows[i - 1][1]) / 1e3)
ks = [r for r in rows if r[2].startswith('kstar_kernel')]
per = [(ks[i + 1][0] - ks[i][0]) / 1e3 for i in range(len(ks) - 1)]
per = [p for p in per if p < 200]
import statistics as st
print('lock-steps: %d, median period %.1f us' % (len(per), st.median(per)))
for n in names:
    if dur[n]:
        print('%-22s dur %.1f us   idle before it %.1f us (median)' % (n, st.median(dur[n]), st.median(gap[n])))
