"""Coarse timeline of the production schedule from a rocprofv3 kernel trace: one row per `step_us` of the last iteration-and-a-bit:
summed chip fill (see trace_fill.py) and, per queue, the kernel that ran longest in the bucket.
usage: python scratch/trace_timeline.py <kernel_trace.csv> [window_ms] [step_us]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 22.0
step = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 100e3


def wgs(r):
    n = 1
    for a, b in (('Grid_Size_X', 'Workgroup_Size_X'), ('Grid_Size_Y', 'Workgroup_Size_Y'), ('Grid_Size_Z', 'Workgroup_Size_Z')):
        g, w = int(r.get(a, 1) or 1), int(r.get(b, 1) or 1)
        n *= max(1, (g + w - 1) // max(w, 1))
    return n


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').replace('gcc_igemm::', '')
    m = re.match(r'([A-Za-z_0-9:]+(<[^(]*>)?)', n)
    n = (m.group(1) if m else n)
    n = n.replace('_kernel', '').replace(', true, true', '').replace(', true, false', '*')
    return n[:22]


ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), min(1.0, wgs(r) / 256.0), short(r['Kernel_Name']), r.get('Queue_Id', '0')) for r in rows]
t1 = max(e[1] for e in ev)
t0 = t1 - win * 1e6
ev = [e for e in ev if e[1] >= t0]
queues = sorted({e[4] for e in ev})
nb = int(win * 1e6 / step)
fill = [0.0] * nb
names = [defaultdict(lambda: defaultdict(float)) for _ in range(nb)]
for s, e, f, n, q in ev:
    b0, b1 = max(0, int((s - t0) / step)), min(nb - 1, int((e - t0) / step))
    for b in range(b0, b1 + 1):
        lo, hi = max(s, t0 + b * step), min(e, t0 + (b + 1) * step)
        if hi > lo:
            fill[b] += f * (hi - lo) / step
            names[b][q][n] += hi - lo
print('  t(ms)  fill  ' + '  '.join('q%-21s' % q for q in queues))
for b in range(nb):
    cols = []
    for q in queues:
        d = names[b].get(q)
        if d:
            n, t = max(d.items(), key=lambda x: x[1])
            cols.append('%-18s %3d%%' % (n[:18], min(100, int(100 * sum(d.values()) / step))))
        else:
            cols.append(' ' * 23)
    print('%7.2f  %4.2f  %s' % (b * step / 1e6, min(fill[b], 9.99), '  '.join(cols)))
