"""Timeline of one bench step from a rocprofv3 --kernel-trace CSV: for the last full step before the trace ends, every
kernel's start (us from the step's first kernel), duration and the idle gap on the device before it starts (no kernel of
any stream running).  Shows what the host puts between the stages of a step (launch latency, a mid-step synchronisation).
    python tools/step_gaps.py <dir with *_kernel_trace.csv> [step index from the end, default 2]"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), (re.search(r'k_\w+', r['Kernel_Name']) or re.search(r'\w+', r['Kernel_Name'])).group(0)[:48]))
rows.sort()
# a step starts with the first PM kernel after a routing kernel
routes = [i for i, r in enumerate(rows) if r[2] == 'k_mrtm_wave']
if len(routes) < back + 1:
    sys.exit('too few routing launches in ' + f)
lo, hi = routes[-back - 1] + 1, routes[-back]
step = rows[lo:hi + 1]
t0 = step[0][0]
busy_until = t0
print('{:>10} {:>10} {:>9}  kernel'.format('start us', 'dur us', 'idle us'))
idle_total = 0.0
for s, e, name in step:
    idle = max(0, s - busy_until) / 1e3
    idle_total += idle
    print('{:10.1f} {:10.1f} {:9.1f}  {}'.format((s - t0) / 1e3, (e - s) / 1e3, idle, name))
    busy_until = max(busy_until, e)
print('step, first kernel start to last kernel end: {:.1f} us; device idle inside it: {:.1f} us'.format((busy_until - t0) / 1e3, idle_total))
nxt = rows[hi + 1][0] if hi + 1 < len(rows) else None
if nxt:
    print('gap to the next step\'s first kernel: {:.1f} us'.format((nxt - busy_until) / 1e3))
