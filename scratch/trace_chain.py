"""Kernel-by-kernel listing of the last whole iteration of a rocprofv3 kernel trace, per queue: start (us from the iteration's
first kernel), duration, gap since the previous kernel of the same queue ended, workgroups, name.  Plus, per queue, the sums:
busy time, gap time, launches; and the gap histogram.
usage: python scratch/trace_chain.py <kernel_trace.csv> [out_listing.txt]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))


def wgs(r):
    n = 1
    for a, b in (('Grid_Size_X', 'Workgroup_Size_X'), ('Grid_Size_Y', 'Workgroup_Size_Y'), ('Grid_Size_Z', 'Workgroup_Size_Z')):
        g, w = int(r.get(a, 1) or 1), int(r.get(b, 1) or 1)
        n *= max(1, (g + w - 1) // max(w, 1))
    return n


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').replace('gcc_igemm::', '')
    m = re.match(r'([A-Za-z_0-9:]+(<[^(]*>)?)', n)
    return (m.group(1) if m else n)[:48]


ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), wgs(r), short(r['Kernel_Name']), r.get('Queue_Id', '0')) for r in rows)
marks = [e[0] for e in ev if e[3].startswith('arch_coeffs')]
if len(marks) < 3:
    raise SystemExit('need >= 3 arch_coeffs launches (whole iterations) in the trace')
t0, t1 = marks[-3], marks[-2]          # a whole iteration away from the end-of-run tail
it = [e for e in ev if t0 <= e[0] < t1]
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else None
print('iteration: %.2f ms, %d launches' % ((t1 - t0) / 1e6, len(it)))
byq = defaultdict(list)
for e in it:
    byq[e[4]].append(e)
for q, lst in sorted(byq.items()):
    busy = sum(e[1] - e[0] for e in lst)
    gaps = [max(0, b[0] - a[1]) for a, b in zip(lst[:-1], lst[1:])]
    small = [g for g in gaps if g < 50e3]
    print('queue %s: %4d launches, busy %.2f ms, gaps < 50 us: %d summing %.2f ms (median %.1f us), longer gaps %d summing %.2f ms' % (
        q, len(lst), busy / 1e6, len(small), sum(small) / 1e6, sorted(small)[len(small) // 2] / 1e3 if small else 0,
        len(gaps) - len(small), (sum(gaps) - sum(small)) / 1e6))
    if out:
        out.write('==== queue %s\n' % q)
        prev = None
        for s, e, w, n, _ in lst:
            out.write('%9.1f us  dur %7.1f  gap %7.1f  wgs %6d  %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, w, n))
            prev = e
